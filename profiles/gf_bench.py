"""Guided-filter kernel variants side by side: time per launch and max |t - t_ref| against the exact-order kernel.

usage: python profiles/gf_bench.py [H W B [k eps]]   (default 2160 3840 16 15 0.5)
Variants are selected through the context's route selectors (uwie_set_tuning: gf_pipe, gf_split) and the mode argument.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import underwater_image_enhancement_amd as uw  # noqa: E402

H, W, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (2160, 3840, 16)))
k = int(sys.argv[4]) if len(sys.argv) > 4 else 15
eps = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
dev = uw.Device(0)
g = torch.Generator(device="cuda").manual_seed(1)
yy = torch.arange(H, device="cuda").view(1, H, 1)
xx = torch.arange(W, device="cuda").view(1, 1, W)
field = 0.5 + 0.25 * torch.sin(xx / 97.0) * torch.cos(yy / 61.0)
gray = (255 * (field + 0.03 * torch.randn((B, H, W), device="cuda", generator=g)).clamp(0, 1)).to(torch.uint8).contiguous()
t0 = (1.0 - 0.5 * (field * 0.9 + 0.05 * torch.rand((B, H, W), device="cuda", generator=g))).clamp(0.1, 1.0).float().contiguous()


def run(sel, exact=False, reps=5):
    dev.tune(gf_pipe=1, gf_split=1, gf_bands=0)
    dev.tune(**sel)
    t = dev.guided_filter(gray, t0, k, eps, exact=exact)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        t = dev.guided_filter(gray, t0, k, eps, exact=exact)
    e1.record()
    torch.cuda.synchronize()
    return t, e0.elapsed_time(e1) / reps


nb = min(B, 2)
ref, ms_ref = run({}, exact=True, reps=1)
ref = ref[:nb].clone()
print(f"{H}x{W} x{B} k={k} eps={eps}")
print(f"  exact-order kernels      {ms_ref:8.3f} ms")
for name, env, mode in (("LDS-tiled strip kernel", {"gf_pipe": 0}, False), ("pipe, f64 ring in LDS", {"gf_split": 0}, False),
                        ("pipe, f64 split ring", {}, False), ("pipe, fx32 ring", {}, 2)):
    t, ms = run(env, exact=mode)
    err = (t[:nb] - ref).abs().max().item()
    gbs = B * H * W * 13 / ms / 1e6
    print(f"  {name:24s} {ms:8.3f} ms   {gbs:7.1f} GB/s algorithmic   max|t - t_exact| = {err:.3e}")
