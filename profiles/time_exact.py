"""Per-kernel times of the exact-order guided filter (uwie_params.gf_exact = 1: cv2.boxFilter's running sums, S6:31-45).
usage: python profiles/time_exact.py [H W B]   (default 2160 3840 16; x4 for the 4K x 64 figure)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import underwater_image_enhancement_amd._lib as _L  # noqa: E402

if os.environ.get("UWIE_AB_LIB"):  # an A/B build (profiles/ab.sh): lib_b or lib_c
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.__file__), os.environ["UWIE_AB_LIB"], "libuwie.so")
import underwater_image_enhancement_amd as uw  # noqa: E402

H, W, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (2160, 3840, 16)))
dev = uw.Device(0)
g = torch.Generator(device="cuda").manual_seed(1)
yy = torch.arange(H, device="cuda").view(1, H, 1)
xx = torch.arange(W, device="cuda").view(1, 1, W)
field = 0.5 + 0.25 * torch.sin(xx / 97.0) * torch.cos(yy / 61.0)
gray = (255 * (field + 0.03 * torch.randn((B, H, W), device="cuda", generator=g)).clamp(0, 1)).to(torch.uint8).contiguous()
t0 = (1.0 - 0.5 * (field * 0.9 + 0.05 * torch.rand((B, H, W), device="cuda", generator=g))).clamp(0.1, 1.0).float().contiguous()
ref = dev.guided_filter(gray, t0, 15, 0.5, exact=True)
torch.cuda.synchronize()
dev.profile(True)
for _ in range(3):
    t = dev.guided_filter(gray, t0, 15, 0.5, exact=True)
rows = dev.profile_rows()
dev.profile(False)
tot = 0.0
for name, (ms, calls) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"  {ms / 3:8.3f} ms  {calls // 3} launches  {name}")
    tot += ms / 3
print(f"exact-order guided filter {H}x{W} x{B}: {tot:.3f} ms per call  ({tot * 64 / B:.1f} ms scaled to x64); checksum {float(t.double().sum()):.12e} identical to first call: {bool(torch.equal(t, ref))}")
