#!/usr/bin/env python3
"""What an fp16 transmission plane would cost in bytes (BASELINE.json configs[4] "fp16 intermediates"):
the CPU oracle with the refined transmission (six_stadigy.py:180) rounded to float16 -- and, for comparison, to float32,
which is what UWIE_INTER_F32T stores -- against the float64 reference path, on the bench's frame classes.  CPU only.

    python3 profiles/fp16_t_experiment.py > profiles/r04_fp16_t_experiment.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import uwie_oracle as orc  # noqa: E402


def underwater(rng, H, W, gains):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    f = 0.5 + 0.25 * (np.sin(xx / 61.0) * np.cos(yy / 47.0) + 0.5 * np.sin((xx + 2 * yy) / 113.0) + 0.3 * np.cos(xx / 29.0) + 0.2 * np.sin(yy / 17.0)) / 2.0
    img = f[:, :, None] * np.array(gains) + rng.normal(0, 0.02, (H, W, 3))
    return np.clip(np.floor(255 * img), 0, 255).astype(np.uint8)


def run(u8, k, dtype):
    S = orc.SixStrategyOracle
    orig = S.transmission.__func__

    def patched(cls, img, A, omega, ksize, eps):
        t = orig(cls, img, A, omega, ksize, eps)
        return t if dtype is None else t.astype(dtype).astype(np.float64)

    S.transmission = classmethod(patched)
    try:
        return orc.enhance_u8(u8, k)
    finally:
        S.transmission = classmethod(orig)


def main():
    rng = np.random.default_rng(16)
    frames = {"underwater": underwater(rng, 480, 640, (0.45, 0.85, 0.80)),
              "hazy": np.floor(255 * (rng.random((480, 640, 3)) * 0.7 + 0.15)).astype(np.uint8),
              "uniform": rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)}
    print("# transmission rounded to the given type before restore_image (S6:183-188); everything else float64 / float32 as the reference")
    print("# frame       strategy  type     bytes differing   > 1 LSB     worst   PSNR dB")
    for name, u8 in frames.items():
        for k in (1, 2, 3):
            ref = run(u8, k, None)
            for tag, dt in (("float32", np.float32), ("float16", np.float16)):
                d = np.abs(run(u8, k, dt).astype(int) - ref.astype(int))
                mse = float((d.astype(np.float64) ** 2).mean())
                psnr = 10 * np.log10(255.0 ** 2 / mse) if mse else float("inf")
                print(f"{name:12s} {k:8d}  {tag:8s} {np.count_nonzero(d) / d.size:14.5f} {np.count_nonzero(d > 1) / d.size:10.5f} {int(d.max()):8d} {psnr:9.1f}")


if __name__ == "__main__":
    main()
