"""Cast-detection check on a few shapes: device mean (sequential float32 emulation) against NumPy's.  python profiles/dbg_cast.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from underwater_image_enhancement_amd.runtime import Device

dev = Device(0)
rng = np.random.default_rng(1)
cases = {
    "white_4x4": np.full((4, 4, 3), 255, np.uint8),
    "white_40x40": np.full((40, 40, 3), 255, np.uint8),
    "noise_8x8": rng.integers(0, 256, (8, 8, 3), dtype=np.uint8),
    "noise_61x83": rng.integers(0, 256, (61, 83, 3), dtype=np.uint8),
    "noise_300x300": rng.integers(0, 256, (300, 300, 3), dtype=np.uint8),
    "noise_1080p": rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8),
}
for name, u8 in cases.items():
    x = u8.astype(np.float32) / np.float32(255.0)
    want = x.mean(axis=(0, 1))
    kind, mean = dev.cast_classify(dev.tensor(u8[None]))
    got = mean.cpu().numpy()[0]
    print(name, "OK" if np.array_equal(got, want) else "DIFF", got, want, flush=True)
