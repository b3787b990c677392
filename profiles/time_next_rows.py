"""Timings of the SURVEY 8f entry points (features, quality scores, differentiable enhancement, six-strategy fan-out) at
4K x 16.  Run on the GPU box from the repo root: python profiles/time_next_rows.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import underwater_image_enhancement_amd as uw
import bench
dev = uw.get_device(0)
B,H,W = 16,2160,3840
fr = bench.synth_frames('underwater', B, H, W, dev.torch_device, 0)
def timeit(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    dev.profile(True); fn(); rows=dev.profile_rows(); dev.profile(False)
    t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    ms=(time.time()-t)/n*1e3
    top=sorted(rows.items(), key=lambda kv:-kv[1][0])[:6]
    print(name, round(ms,2), 'ms;', ', '.join(f'{k}={v[0]:.2f}' for k,v in top))
timeit('features', lambda: dev.extract_features_u8(fr))
timeit('quality_u8', lambda: dev.quality_scores(fr))
f32 = (fr.float()/255.0).contiguous()
timeit('quality_u8+f32', lambda: dev.quality_scores(fr, f32))
params = torch.tensor([[5.0, 95.0, 0.5, 1.2]]*B, dtype=torch.float32, device=dev.torch_device)
timeit('diff_enhance_hwc', lambda: dev.diff_enhance_f32(f32, params, planar=False))
pl = f32.permute(0,3,1,2).contiguous()
timeit('diff_enhance_planar', lambda: dev.diff_enhance_f32(pl, params, planar=True))
timeit('enhance_all', lambda: dev.enhance_all_u8(fr))
