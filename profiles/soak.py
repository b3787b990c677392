"""Long randomised differential run of uw.enhance against the oracle (same generator as tests/test_gpu_fuzz.py, more frames,
larger sizes, a different seed per run):   python profiles/soak.py [seed] [frames] [max_side]
Prints one line per mismatching case and a summary; exit code 1 when a byte is off by more than 1 LSB.
SOAK_GF_EXACT=1: the three dehazing strategies only, through the exact-order guided filter (uwie_params.gf_exact = 1).
SOAK_ENTRY=1 (round 4): every frame is large enough for a launched quadtree level and has a width that is a multiple of 8, a
quarter of them with a colour cast near detect_image_type's threshold -- the frames the fused entry pass takes (tuning
entry_fuse: level-0 histograms and the gray plane out of cast detection's chunk pass); the three dehazing strategies."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import underwater_image_enhancement_amd as uw  # noqa: E402
from oracle import uwie_oracle as orc  # noqa: E402
from test_gpu_fuzz import random_frame  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rng = np.random.default_rng(seed)
    exact = os.environ.get("SOAK_GF_EXACT") == "1"
    entry = os.environ.get("SOAK_ENTRY") == "1"
    t0 = time.time()
    cases = differing = bad = 0
    for i in range(frames):
        u8 = random_frame(rng)
        if max_side and i % 4 == 0:  # a larger frame now and then: several histogram chunks, launched quadtree levels
            H, W = int(rng.integers(200, max_side)), int(rng.integers(200, max_side))
            u8 = np.ascontiguousarray(np.resize(np.tile(u8, (H // u8.shape[0] + 1, W // u8.shape[1] + 1, 1))[:H, :W], (H, W, 3)))
            u8 = np.clip(u8.astype(int) + rng.integers(-3, 4, u8.shape), 0, 255).astype(np.uint8)
        if entry:
            H, W = int(rng.integers(140, max(max_side, 400))), 8 * int(rng.integers(30, max(max_side, 400) // 8))
            u8 = np.ascontiguousarray(np.tile(u8, (H // u8.shape[0] + 1, W // u8.shape[1] + 1, 1))[:H, :W])
            u8 = np.clip(u8.astype(int) + rng.integers(-3, 4, u8.shape), 0, 255).astype(np.uint8)
            if i % 4 == 1:  # means that differ by about the 0.05 of detect_image_type: both decisions, and wrong guesses, occur
                m = u8.reshape(-1, 3).mean(axis=0)
                ch = 1 + (i // 4) % 2
                shift = (m[0] - m[ch]) + 255 * (0.05 + rng.normal(0, 0.004))
                u8[:, :, ch] = np.clip(u8[:, :, ch].astype(float) + shift, 0, 255).astype(np.uint8)
        for k in ((1, 2, 3) if (exact or entry) else (1, 2, 3, 4, 5, 6)):
            got, want = (uw.enhance(u8, strategy=k, gf_exact=1) if exact else uw.enhance(u8, strategy=k)), orc.enhance_u8(u8, k)
            d = np.abs(got.astype(int) - want.astype(int))
            cases += 1
            n = int(np.count_nonzero(d))
            if n:
                differing += n
                print(f"frame {i} {u8.shape} strategy {k}: {n} bytes differ, max {d.max()} LSB", flush=True)
                bad += int(d.max() > 1)
        if i % 20 == 19:
            print(f"... {i + 1} frames, {cases} cases, {differing} differing bytes, {time.time() - t0:.0f} s", flush=True)
    print(f"seed {seed}: {cases} cases, {differing} differing bytes, {bad} cases beyond 1 LSB, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
