"""Times the quadtree (uwie_atmospheric_light) alone: python profiles/q_bench.py [H W B].
With the library built with -DUWIE_TAIL_PROF the trace's score fields of the k_q_tail levels hold wavefront 0's phase
times in 10 ns ticks (sums, gray load, Canny) and its edge count."""
import sys, time
import torch
sys.path.insert(0, ".")
from underwater_image_enhancement_amd import runtime as rt_mod
import bench

def main():
    H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1080, 1920, 1)
    rt = rt_mod.get_device(0)
    frames = bench.synth_frames("underwater", B, H, W, "cuda", 1)
    kind = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(3):
        A = rt.atmospheric_light(frames, kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        A = rt.atmospheric_light(frames, kind)
    torch.cuda.synchronize()
    print(f"{H}x{W} batch {B}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call")
    A, tr = rt.atmospheric_light(frames, kind, trace=True)
    for lv in range(14):
        r = tr[0, lv]
        print(lv, r["y0"], r["x0"], r["rows"], r["cols"], r["score"])

main()
