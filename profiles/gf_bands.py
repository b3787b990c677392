"""The split guided kernel at 4K x 64 with forced band counts (tuning gf_bands): python profiles/gf_bands.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import underwater_image_enhancement_amd as uw  # noqa: E402

H, W = 2160, 3840
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = uw.Device(0)
g = torch.Generator(device="cuda").manual_seed(1)
yy = torch.arange(H, device="cuda").view(1, H, 1)
xx = torch.arange(W, device="cuda").view(1, 1, W)
field = 0.5 + 0.25 * torch.sin(xx / 97.0) * torch.cos(yy / 61.0)
gray = (255 * (field + 0.03 * torch.randn((B, H, W), device="cuda", generator=g)).clamp(0, 1)).to(torch.uint8).contiguous()
t0 = (1.0 - 0.5 * (field * 0.9 + 0.05 * torch.rand((B, H, W), device="cuda", generator=g))).clamp(0.1, 1.0).float().contiguous()
for rnd in range(2):
    for bands in (0, 2, 3, 4, 5, 6, 7, 8, 10, 12):
        dev.tune(gf_pipe=1, gf_split=1, gf_bands=bands)
        dev.guided_filter(gray, t0, 15, 1e-3, exact=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dev.guided_filter(gray, t0, 15, 1e-3, exact=False)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} gf_bands={bands:2d}: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
