"""One-off randomised check on the GPU box: random frames x random strategy (both surfaces) x random route-selector set against the
oracle; prints every case that differs.  SEED / N from the environment.  python profiles/fuzz_knobs.py"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import underwater_image_enhancement_amd as uw
from oracle import uwie_oracle as orc
from test_gpu_fuzz import random_frame
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
KNOBS = [{}, {"restore_store": 1}, {"lin_no_predict": 1}, {"lin_predict_shift": 3}, {"lin_cap": 24},
         {"lin_predict_shift": 300, "restore_store": 1}, {"select_generic": 1}, {"streams": 2}]  # uwie_set_tuning selectors
DEFAULTS = {"restore_store": 0, "lin_no_predict": 0, "lin_predict_shift": 0, "lin_cap": 0, "select_generic": 0, "streams": 1}
ES = orc.DictStrategyOracle
names = ["strong_dehazing", "medium_dehazing", "light_enhancement", "clahe_enhancement", "histogram_equalization"]
bad = tot = 0
for i in range(int(os.environ.get("N", "150"))):
    u8 = random_frame(rng)
    knobs = KNOBS[rng.integers(len(KNOBS))]
    uw.get_device(0).tune(**knobs)
    try:
        if rng.random() < float(os.environ.get("FUZZ_SIX_SHARE", "0.7")):
            k = int(rng.integers(1, 7))
            got, want = uw.enhance(u8, strategy=k), orc.enhance_u8(u8, k)
            what = f"six {k}"
        else:
            name = names[rng.integers(5)]
            x = orc.normalise_u8(u8)
            # the dict surface returns float64 like the reference: compared bit for bit (equal without gamma, as in the tests)
            want = ES.run(x, name, {}).view(np.uint64)
            if os.environ.get("FUZZ_GF_EXACT") == "1":  # cv2.boxFilter's summation order: no tolerance on t left
                from underwater_image_enhancement_amd import _lib
                dev = uw.get_device(0)
                p = dev.params(_lib.SURFACE_DICT, _lib.DICT_STRATEGIES[name], gf_exact=1, apply_gamma=0)
                got = dev.enhance_u8_f64(dev.tensor(u8[None]), p)[1][0].cpu().numpy().view(np.uint64)
            else:
                got = np.ascontiguousarray(uw.EnhancementStrategies.apply_strategy(x, name, {})).view(np.uint64)
            what = name
    finally:
        uw.get_device(0).tune(**DEFAULTS)
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    tot += 1
    if d.max() > 0:
        bad += 1
        print("DIFF", i, u8.shape, what, knobs, "max", d.max(), "count", int((d > 0).sum()))
print("cases", tot, "with differences", bad)
