"""Quadtree routes against each other on random frames (tuning q_hist = 1 / 0): the atmospheric light must be identical.
python profiles/soak_quadtree.py [seed] [frames]   -- frames between 200 and 1400 px a side so that 1 .. 3 levels are launched"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import underwater_image_enhancement_amd as uw  # noqa: E402
from test_gpu_fuzz import random_frame  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
dev = uw.get_device(0)
t0 = time.time()
bad = 0
for i in range(frames):
    u8 = random_frame(rng)
    H, W = int(rng.integers(200, 1400)), int(rng.integers(200, 1400))
    u8 = np.ascontiguousarray(np.tile(u8, (H // u8.shape[0] + 1, W // u8.shape[1] + 1, 1))[:H, :W])
    if i % 3:
        u8 = np.clip(u8.astype(int) + rng.integers(-2, 3, u8.shape), 0, 255).astype(np.uint8)  # (i % 3 == 0: exact repeats -> ties)
    kk = torch.tensor([int(rng.integers(0, 3))], dtype=torch.int32, device=dev.torch_device)
    res = []
    for q in (1, 0):
        with dev.tuning(q_hist=q):
            res.append(dev.atmospheric_light(dev.tensor(u8[None]), kk).cpu().numpy())
    if not np.array_equal(res[0], res[1]):
        bad += 1
        print(f"frame {i} {u8.shape} kind {int(kk[0])}: {res[0]} != {res[1]}", flush=True)
    if i % 50 == 49:
        print(f"... {i + 1} frames, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"seed {seed}: {frames} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
