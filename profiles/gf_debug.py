import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import underwater_image_enhancement_amd as uw
H, W, B = (int(v) for v in sys.argv[1:4])
k = 15; eps = 0.5
dev = uw.Device(0)
g = torch.Generator(device="cuda").manual_seed(1)
yy = torch.arange(H, device="cuda").view(1, H, 1); xx = torch.arange(W, device="cuda").view(1, 1, W)
field = 0.5 + 0.25 * torch.sin(xx / 97.0) * torch.cos(yy / 61.0)
gray = (255 * (field + 0.03 * torch.randn((B, H, W), device="cuda", generator=g)).clamp(0, 1)).to(torch.uint8).contiguous()
t0 = (1.0 - 0.5 * (field * 0.9 + 0.05 * torch.rand((B, H, W), device="cuda", generator=g))).clamp(0.1, 1.0).float().contiguous()
os.environ["UWIE_GF_RING_FORCE"] = "0"
ref = dev.guided_filter(gray, t0, k, eps, exact=False).clone()
os.environ["UWIE_GF_RING_FORCE"] = "1"
for rep in range(3):
    t = dev.guided_filter(gray, t0, k, eps, exact=False)
    torch.cuda.synchronize()
    bad = (t - ref).abs() > 1e-9
    nb = int(bad.sum())
    print("rep", rep, "bad", nb)
    if nb:
        idx = bad.nonzero()
        print(" frames", torch.unique(idx[:, 0]).tolist()[:20])
        print(" rows", int(idx[:, 1].min()), int(idx[:, 1].max()), "cols", int(idx[:, 2].min()), int(idx[:, 2].max()))
        rows = torch.unique(idx[:, 1]).tolist(); print(" nrows", len(rows), rows[:40])
        cols = torch.unique(idx[:, 2]).tolist(); print(" ncols", len(cols), cols[:40])
        print(" sample vals", t[bad][:8].tolist())
