#!/usr/bin/env python3
"""Output hash of one library build on seeded frames, for A/B builds that must give identical bytes:
    python3 profiles/ab_equal.py lib_b [H W B strategy]      (prints sha256 of the u8 output; compare across builds)
Each build runs in its own process (the library path is fixed at import)."""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/profiles/", 1)[0])
import underwater_image_enhancement_amd._lib as L  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 else "lib"
H, W, B, K = (int(x) for x in (sys.argv[2:6] if len(sys.argv) >= 6 else (1080, 1920, 48, 2)))
L.LIB_PATH = L.LIB_PATH.replace("/lib/libuwie.so", f"/{lib}/libuwie.so")
import underwater_image_enhancement_amd as uw  # noqa: E402

rng = np.random.default_rng(7)
# smooth underwater-like gradient + noise, different per frame
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
frames = np.empty((B, H, W, 3), np.uint8)
for b in range(B):
    base = np.stack([40 + 30 * np.sin(xx / (90 + b)) + 20 * yy / H, 110 + 50 * np.cos(yy / (70 + b)), 150 + 60 * np.sin((xx + yy) / (120 + 2 * b))], -1)
    frames[b] = np.clip(base + rng.normal(0, 6 + b % 5, base.shape), 0, 255).astype(np.uint8)
dev = uw.get_device()
p = dev.params(0, K)
out = dev.enhance_u8(dev.tensor(frames), p)
out = out[0] if isinstance(out, tuple) else out
torch.cuda.synchronize()
print(lib, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], tuple(out.shape))
