"""Per-strategy timings at 4K x 16 with the six largest kernels of each (both surfaces).  Run on the GPU box from the repo
root: python profiles/time_strategies.py"""
import sys, time
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
for k in (1,2,3,4,5,6):
    p = dev.params(_lib.SURFACE_SIX, k, cast_correct=1)
    dev.profile(True); dev.enhance_u8(fr,p); rows=dev.profile_rows(); dev.profile(False)
    top=sorted(rows.items(), key=lambda kv:-kv[1][0])[:7]
    print('six', k, round(run(p),2), 'ms;', ', '.join(f'{n}={v[0]:.2f}' for n,v in top))
for k in range(5):
    p = dev.params(_lib.SURFACE_DICT, k)
    dev.profile(True); dev.enhance_u8(fr,p); rows=dev.profile_rows(); dev.profile(False)
    top=sorted(rows.items(), key=lambda kv:-kv[1][0])[:6]
    print('dict', k, round(run(p),2), 'ms;', ', '.join(f'{n}={v[0]:.2f}' for n,v in top))
