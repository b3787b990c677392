"""Does the 4K x 64 job run faster as sub-batches whose intermediate planes (t: 66 MB per frame as float64) could stay in the
256 MB Infinity Cache between the kernel that writes them and the two that read them?  The same 64 frames through
uwie_enhance_u8 in calls of c frames each, back to back on one stream.
usage: python profiles/chunk_sweep.py [H W B]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import underwater_image_enhancement_amd as uw  # noqa: E402
from underwater_image_enhancement_amd import _lib  # noqa: E402

H, W, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (2160, 3840, 64)))
dev = uw.get_device(0)
frames = bench.synth_frames("underwater", B, H, W, dev.torch_device, seed=2000)
p = dev.params(_lib.SURFACE_SIX, 2)
out = dev.empty((B, H, W, 3), torch.uint8)
ref = None
for c in (64, 32, 16, 8, 4, 2, 1):
    if c > B:
        continue
    ws = dev.workspace_for(c, H, W, p)

    def run():
        for i in range(0, B, c):
            n = min(c, B - i)
            _lib.check(dev.lib.uwie_enhance_u8(dev._ctx, ctypes.c_void_p(frames[i:].data_ptr()), ctypes.c_void_p(out[i:].data_ptr()),
                                               None, n, H, W, ctypes.byref(p), ctypes.c_void_p(ws.data_ptr()), ws.numel(), dev.stream()))

    run()
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"sub-batch {c:3d} frames: {ms:8.3f} ms per {B} frames  ({B * H * W / ms / 1e6:7.2f} GP/s)  identical to one call: {same}", flush=True)
    del ws
