"""Round-3 debugging aid: which route of the dehazing tail disagrees with the oracle on the test frames."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import underwater_image_enhancement_amd as uw
from oracle import uwie_oracle as orc
from test_gpu_stages import frames_for_tests

frames = frames_for_tests(np.random.default_rng(4242))
frames.pop("tiny_5x7")
for k in (1, 2, 3):
    for name, u8 in frames.items():
        want = orc.enhance_u8(u8, k)
        row = []
        for env in ({}, {"restore_store": 1}, {"select_generic": 1}, {"lin_no_predict": 1}):
            with uw.get_device().tuning(**env):
                got = uw.enhance(u8, strategy=k)
            d = np.abs(got.astype(int) - want.astype(int))
            row.append(f"{list(env) or 'default'}: max {d.max()} n {np.count_nonzero(d)}")
        print(k, name, u8.shape, " | ".join(row))
