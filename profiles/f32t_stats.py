"""UWIE_INTER_F32T (float32 transmission, float32 restore) against the oracle: byte statistics and timings.
Run on the GPU box from the repo root: python profiles/f32t_stats.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import underwater_image_enhancement_amd as uw
from underwater_image_enhancement_amd import _lib
from oracle import uwie_oracle as orc
from test_gpu_configs import underwater

rng = np.random.default_rng(7)
frames = [("uw_480x640", underwater(rng, 480, 640, (0.45, 0.85, 0.80))), ("uw_600x800b", underwater(rng, 600, 800, (0.45, 0.75, 0.90))),
          ("noise_480x640", rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)),
          ("hazy_480x640", np.floor(255 * (rng.random((480, 640, 3)) * 0.7 + 0.15)).astype(np.uint8)),
          ("uw_1080p", underwater(rng, 1080, 1920, (0.45, 0.85, 0.80)))]
for k in (1, 2, 3):
    tot = diff = beyond = 0
    worst = 0
    mse = 0.0
    for name, u8 in frames:
        want = orc.enhance_u8(u8, k)
        got = uw.enhance(u8, strategy=k, inter_dtype=_lib.INTER_F32T)
        d = np.abs(got.astype(int) - want.astype(int))
        tot += d.size; diff += int(np.count_nonzero(d)); beyond += int(np.count_nonzero(d > 1)); worst = max(worst, int(d.max()))
        mse += float((d.astype(np.float64) ** 2).sum())
        print(f"  strategy {k} {name}: {np.count_nonzero(d)} of {d.size} differ, {np.count_nonzero(d > 1)} by more than 1, max {d.max()}")
    psnr = 10 * np.log10(255.0 ** 2 / (mse / tot)) if mse else float("inf")
    print(f"strategy {k}: {diff / tot:.2e} of bytes differ, {beyond / tot:.2e} by more than 1 LSB, worst {worst}, PSNR {psnr:.1f} dB")

dev = uw.get_device(0)
import bench
B, H, W = 64, 2160, 3840
fr = bench.synth_frames("underwater", B, H, W, dev.torch_device, 0)
for mode in (0, 2):
    p = dev.params(_lib.SURFACE_SIX, 2, inter_dtype=mode)
    dev.enhance_u8(fr, p); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        dev.enhance_u8(fr, p)
    torch.cuda.synchronize()
    ms = (time.time() - t) / 5 * 1e3
    dev.profile(True); dev.enhance_u8(fr, p); rows = dev.profile_rows(); dev.profile(False)
    pick = {n: round(v[0], 3) for n, v in rows.items() if n in ("k_guided_split", "k_restore_hist_collect", "k_stretch_lab_lut<1>", "k_lin_sample<float>")}
    print(f"inter_dtype={mode}: {ms:.2f} ms per 4K x 64 step; {pick}")
