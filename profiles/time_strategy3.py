import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import underwater_image_enhancement_amd as uw
from underwater_image_enhancement_amd import _lib
import bench
dev = uw.get_device(0)
B,H,W = 16,2160,3840
fr = bench.synth_frames('underwater', B, H, W, dev.torch_device, 0)
def run(p, n=3):
    dev.enhance_u8(fr, p); torch.cuda.synchronize()
    t=time.time()
    for _ in range(n): dev.enhance_u8(fr, p)
    torch.cuda.synchronize()
    return (time.time()-t)/n*1e3
for env in ({"lin_predict3": 0}, {"lin_predict3": 1}):
    dev.tune(**env)
    p = dev.params(_lib.SURFACE_SIX, 3, cast_correct=1)
    dev.profile(True); dev.enhance_u8(fr,p); rows=dev.profile_rows(); dev.profile(False)
    top=sorted(rows.items(), key=lambda kv:-kv[1][0])[:8]
    print(env, round(run(p),2), 'ms;', ', '.join(f'{n}={v[0]:.2f}' for n,v in top))
