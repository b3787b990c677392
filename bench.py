"""Throughput of the canonical enhance() on synthetic frames resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One step = one pass of uwie_enhance_u8 (strategy 2, cast correction on: the canonical path of SURVEY.md section 8)
over one batch of synthetic "underwater" uint8 frames per GPU.  Default workload: BASELINE.json configs[2], the
configuration the target metric is quoted on -- 4K (3840x2160) RGB, batch 64 per GPU.  Batches shard across ranks
with no data-path collective (frames are independent), so scaling is weak; the only collectives are the barriers
and the MAX of the elapsed time.  Prints ONE JSON line on rank 0.

`roofline`    : the dominant kernel of the timed region (HIP events on the launch stream, recorded inside libuwie),
                its algorithmic bytes (table below) over its measured time, against 8 TB/s HBM.
`cpu_baseline`: the CPU oracle (oracle/uwie_oracle.py: NumPy + C restatement, single thread, kind "port") timed on
                rank 0 at N=1 on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PIPELINE_BYTES_PER_PX = 19.0  # five dependent streaming phases over the u8 frame + 3 B written

# Algorithmic bytes per frame pixel moved by ALL launches of a kernel in one step (per launch = this / launches).
# Quadtree kernels run once per level on a block a quarter the size of the previous one: sum = 4/3.
Q = 4.0 / 3.0
KERNEL_BYTES_PER_PX = {
    "k_chunk_hist": 3, "k_chunk_hist_quad": 3 + 1, "k_quad_hist_reduce": 0, "k_kind_guess": 0, "k_gray_strong": 3 + 1, "k_chunk_ulps": 0, "k_cast_resolve": 0, "k_cast_decide": 0, "k_quant_gray": 3 + 1,
    "k_q_hist_gray": 3 + 1, "k_q_hist_gray_narrow": 3 + 1, "k_q_hist_wide": 3 * (Q - 1), "k_q_hist_narrow": 3 * (Q - 1), "k_q_decide": 0,
    "k_q_chunk_sums<false>": 3 * Q, "k_q_chunk_sums<true>": 3 * Q, "k_canny_gradnms<true>": 1 * Q, "k_canny_gradnms_weak": (1 + 1) * Q,
    "k_canny_union<true>": 1 * Q, "k_canny_mark<true>": 1 * Q, "k_canny_emit<true>": 1 * Q,
    "k_trans_init": 3 + 4, "k_guided_fast<TH>": 1 + 4 + 8,
    # the default guided filter is two launches that split the rows of a frame (uwie_guided_plan): main() scales these by
    # the fraction of the rows each covers
    "k_guided_split": 1 + 4 + 8, "k_guided_pipe": 1 + 4 + 8,
    # round 4: the guided filter with the transmission's first half fused in (reads the u8 frame instead of a float32 t0 plane)
    "k_guided_split8": 3 + 1 + 8,
    "k_box_rows<SrcGuide>": 1 + 4 + 4 * 8, "k_box_cols<EpiAB>": 4 * 8 + 2 * 8,
    "k_box_rows<SrcPlanes2>": 2 * 8 + 2 * 8, "k_box_cols<EpiQ>": 2 * 8 + 1 + 8, "k_restore": 3 + 8 + 12,
    "k_sel_hist<V>": 2 * 3 * 4, "k_restore_hist_collect": 3 + 8 + 12, "k_restore_hist_lin": 3 + 8 + 12,
    "k_restore_hist_key": 3 + 8 + 12, "k_restore_rank": 3 + 8, "k_lin_collect": 12, "k_stretch_lab_lut<1>": 11 + 3, "k_stretch_lab_lut<0>": 12 + 3, "k_clahe_apply_u8": 3 + 3, "k_clahe_apply_f32": 3 + 3 + 12,
    "k_stretch_out": 12 + 3, "k_stretch_apply": 12 + 12, "k_quant_rgb2lab": 12 + 3, "k_clahe_lut": 1,
    "k_clahe_apply": 1 + 1, "k_lab2rgb_f32": 3 + 12, "k_gamma": 12 + 12, "k_quantise": 12 + 3,
    "k_normalise_correct": 3 + 12,
}


def synth_underwater(batch, H, W, device, seed):
    """Seeded synthetic frames: smooth field x greenish/bluish channel gains + noise."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy = torch.arange(H, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    out = torch.empty((batch, H, W, 3), dtype=torch.uint8, device=device)
    for b in range(batch):
        ph = torch.rand(8, generator=g, device=device) * 6.283
        field = 0.55 + 0.25 * (torch.sin(xx / (W / 9.0) + ph[0]) * torch.cos(yy / (H / 7.0) + ph[1])
                               + 0.5 * torch.sin((xx + 2 * yy) / (W / 5.0) + ph[2])
                               + 0.5 * torch.cos((2 * xx - yy) / (W / 13.0) + ph[3])) / 2.0
        gains = (0.45, 0.85, 0.80) if b % 2 == 0 else (0.45, 0.75, 0.90)
        for c in range(3):
            ch = field * gains[c] + torch.randn((H, W), generator=g, device=device) * 0.02
            out[b, :, :, c] = torch.clamp(torch.floor(ch * 255.0), 0, 255).to(torch.uint8)
    return out


def synth_frames(dist, batch, H, W, device, seed):
    """underwater (headline), or the reference's own self-test inputs: uniform bytes (enhancement_strategies.py:516) and
    the hazy range floor(255 * (rand * 0.7 + 0.15)) (example_usage.py:112)."""
    import torch

    if dist == "underwater":
        return synth_underwater(batch, H, W, device, seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand((batch, H, W, 3), generator=g, device=device)
    if dist == "hazy":
        u = u * 0.7 + 0.15
        return torch.floor(u * 255.0).to(torch.uint8)
    return torch.clamp(torch.floor(u * 256.0), 0, 255).to(torch.uint8)


def cpu_baseline(frames_host, gpu_out_host, strategy, dist):
    """The CPU oracle (NumPy + C restatement of the reference path, one thread) on a bounded sample of the workload, as
    BASELINE.md section 3 lays out: the median of five timings at each of 640x480, 1920x1080 and the frame size of the
    timed batch.  At the batch's own size the five runs are its first five frames, and the GPU output of those frames is
    compared with what the oracle returned (max |delta| in LSB, uint8 PSNR).  `value` is the rate at the batch's size."""
    import numpy as np
    import torch

    from oracle import uwie_oracle as orc

    H, W = frames_host.shape[1:3]
    orc.enhance_u8(np.ascontiguousarray(frames_host[0, : H // 8, : W // 8]), strategy)  # warm-up (library load, tables)
    by_size, t_all = {}, time.perf_counter()
    for (h, w) in ((480, 640), (1080, 1920)):
        if (h, w) == (H, W):
            continue
        fr = synth_frames(dist, 5, h, w, torch.device("cpu"), seed=1000 * 1).numpy()
        ts = []
        for i in range(5):
            t1 = time.perf_counter()
            orc.enhance_u8(fr[i], strategy)
            ts.append(time.perf_counter() - t1)
        by_size[f"{w}x{h}"] = round(h * w / 1e6 / sorted(ts)[2], 3)
    ts, worst, sq, nbytes = [], 0, 0.0, 0
    n = min(5, len(frames_host))
    for i in range(n):
        t1 = time.perf_counter()
        want = orc.enhance_u8(frames_host[i], strategy)
        ts.append(time.perf_counter() - t1)
        d = np.abs(want.astype(np.int16) - gpu_out_host[i].astype(np.int16))
        worst = max(worst, int(d.max()))
        sq += float((d.astype(np.float64) ** 2).sum())
        nbytes += d.size
    by_size[f"{W}x{H}"] = round(H * W / 1e6 / sorted(ts)[len(ts) // 2], 3)
    mse = sq / nbytes
    return {"value": by_size[f"{W}x{H}"], "unit": "megapixels/sec", "cores": 1, "kind": "port",
            "sample": f"median of {n} frames of {W}x{H} (the first frames of the timed batch) through oracle.enhance_u8 "
                      f"(NumPy + C restatement of the reference path, 1 thread); 640x480 and 1920x1080: median of 5 frames each; "
                      f"{time.perf_counter() - t_all:.1f} s in all",
            "megapixels_per_sec_by_size": by_size,
            "gpu_vs_oracle_max_lsb": worst, "gpu_vs_oracle_psnr_db": None if mse == 0 else round(10 * math.log10(255.0 ** 2 / mse), 2),
            "gpu_vs_oracle_bytes_compared": nbytes}


TRAFFIC_PROFILE = "profiles/r04_traffic.json"


def measured_traffic(kernel, H, W, B, strategy, launches_per_step, dist="underwater"):
    """HBM bytes per launch of `kernel` from the COMMITTED PMC profile of this same workload (TRAFFIC_PROFILE: FETCH_SIZE x2 +
    WRITE_SIZE, separate rocprofv3 passes of this command, profiles/collect_r03.sh), or None when the workload differs / was
    not profiled.  It is not measured in this run (counters need the profiler); the bench line says so in
    roofline.traffic_source."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    if not (os.path.exists(path) and (H, W, strategy, dist) == (2160, 3840, 2, "underwater")):
        return None
    table = json.load(open(path))["kernels"]
    key = kernel.replace("<TH>", "<8>").replace("<V>", "<float>")
    if key not in table:
        key = key.split("<")[0]
    if key not in table:
        return None
    return table[key]["hbm_bytes_per_px_per_step"] * B * H * W / max(launches_per_step, 1)


def timed_enhance(dev, _lib, torch, frames, strategy, steps, **overrides):
    """ms per call of uwie_enhance_u8 on `frames` (one warm-up, then `steps` calls between two synchronisations)."""
    import ctypes

    B, H, W = frames.shape[:3]
    p = dev.params(_lib.SURFACE_SIX, strategy, **overrides)
    ws = dev.workspace_for(B, H, W, p)
    out = dev.empty((B, H, W, 3), torch.uint8)

    def call():
        _lib.check(dev.lib.uwie_enhance_u8(dev._ctx, ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           None, B, H, W, ctypes.byref(p), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                           dev.stream()))

    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def extras(dev, args, torch, _lib):
    """Numbers SURVEY.md section 8d asks to see beside the headline: the other two input distributions on the same
    workload (uniform noise is the worst case for the Canny / quadtree stages) and BASELINE.json configs[1]."""
    H, W, B = args.height, args.width, args.batch
    res = {}
    for dist in ("uniform", "hazy"):
        if dist == args.dist:
            continue
        fr = synth_frames(dist, B, H, W, dev.torch_device, seed=1000 * 2)
        ms = timed_enhance(dev, _lib, torch, fr, args.strategy, 2)
        res[f"{dist}_megapixels_per_sec"] = round(B * H * W / 1e3 / ms, 1)
        del fr
    # opt-in: sub-batches on two streams (tuning streams = 2), whole-job rate on the headline workload
    fr = synth_frames(args.dist, B, H, W, dev.torch_device, seed=1000 * 2)
    with dev.tuning(streams=2):
        ms = timed_enhance(dev, _lib, torch, fr, args.strategy, 3)
    res["two_streams_megapixels_per_sec"] = round(B * H * W / 1e3 / ms, 1)
    # opt-in: reduced-precision intermediates (BASELINE.json configs[4]; uwie_params.inter_dtype = UWIE_INTER_F32T: float32
    # transmission plane and float32 restore, its own stated tolerance -- tests/test_gpu_fuzz.py); the two sweeps it
    # shortens are reported for both modes (HIP events around the kernels of one recorded step each)
    if args.strategy in (1, 2, 3):
        ms = timed_enhance(dev, _lib, torch, fr, args.strategy, 3, inter_dtype=_lib.INTER_F32T)
        res["f32t_megapixels_per_sec"] = round(B * H * W / 1e3 / ms, 1)
        for tag, mode in (("f64", _lib.INTER_F64), ("f32t", _lib.INTER_F32T)):
            dev.profile(True)
            timed_enhance(dev, _lib, torch, fr, args.strategy, 1, inter_dtype=mode)
            rows = dev.profile_rows()
            dev.profile(False)
            res[f"{tag}_restore_hist_plus_stretch_lab_ms"] = round(
                sum(v[0] / max(v[1], 1) for n, v in rows.items() if n.startswith(("k_restore_hist_collect", "k_restore_rank", "k_stretch_lab_lut"))), 3)
    del fr
    # N1: the batch driver's fan-out, all six strategies per frame with shared cast detection / quadtree
    fan = synth_frames("underwater", min(B, 16), H, W, dev.torch_device, seed=1000 * 2)
    dev.enhance_all_u8(fan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.enhance_all_u8(fan)
    torch.cuda.synchronize()
    res["all_six_strategies_input_megapixels_per_sec"] = round(fan.shape[0] * H * W / 1e6 / (time.perf_counter() - t0), 1)
    # N2's purpose (main.py:118-146): the five Config.STRATEGIES on every frame, comprehensive_assessment of each result, the
    # best one -- one device call (uwie_select_best_u8; input pixels per second, every frame enhanced five times and scored)
    import underwater_image_enhancement_amd as uw
    from underwater_image_enhancement_amd.api import CONFIG_QUALITY_WEIGHTS, CONFIG_STRATEGIES, QUALITY_KEYS, _dict_params

    plist = [_dict_params(dev, k, v) for k, v in CONFIG_STRATEGIES.items()]
    wts = [CONFIG_QUALITY_WEIGHTS.get(k, 0) for k in QUALITY_KEYS]
    dev.select_best_u8(fan, plist, wts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.select_best_u8(fan, plist, wts)
    torch.cuda.synchronize()
    res["select_best_of_five_input_megapixels_per_sec"] = round(fan.shape[0] * H * W / 1e6 / (time.perf_counter() - t0), 1)
    # the rank-counting sweep's two kernels against the histogram sweep's (both selections of one recorded step each)
    if args.strategy in (1, 2):
        fr = synth_frames(args.dist, B, H, W, dev.torch_device, seed=1000 * 2)
        for tag, knob in (("rank", 1), ("hist", 0)):
            with dev.tuning(rank_sweep=knob):
                dev.profile(True)
                timed_enhance(dev, _lib, torch, fr, args.strategy, 1)
                rows = dev.profile_rows()
                dev.profile(False)
            res[f"selection_{tag}_sweep_ms"] = round(sum(v[0] / max(v[1], 1) for n, v in rows.items()
                                                       if n.startswith(("k_restore_rank", "k_restore_hist_collect", "k_lin_", "k_rank_"))), 3)
        del fr
    del fan
    # configs[4] on one GPU: 4K frames from pinned host memory and back, copies overlapped with compute (PCIe-inclusive)
    chunk, nchunks = 8, 8
    se = uw.StreamEnhancer(H, W, chunk=chunk, depth=3, strategy=args.strategy)
    src = synth_frames("underwater", chunk, H, W, dev.torch_device, seed=1000 * 4).cpu()
    for i in range(se.depth):
        se.input_slot(i).copy_(src)
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nchunks):
            if i >= se.depth - 1:
                se.result()
            se.input_slot(i)  # (a real producer would write the next frames here)
            se.submit_slot(i)
        while se._pending:
            se.result()
        dt = time.perf_counter() - t0
    res["configs4_stream_4k_pinned_host_megapixels_per_sec"] = round(nchunks * chunk * H * W / 1e6 / dt, 1)
    del se, src
    one = synth_frames("underwater", 1, 1080, 1920, dev.torch_device, seed=1000 * 1)
    ms = timed_enhance(dev, _lib, torch, one, args.strategy, 10)
    res["configs1_1080p_batch1_ms"] = round(ms, 3)
    res["configs1_1080p_batch1_megapixels_per_sec"] = round(1080 * 1920 / 1e3 / ms, 1)
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this script with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rendezvous on 127.0.0.1.  Returns the worst child exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def main_scatter(args, uw, _lib, torch, world, rank, local, rehearsal):
    """BASELINE.json configs[3]: rank 0 holds world x batch frames in HBM; one step = scatter (point-to-point shards over
    RCCL / xGMI) -> enhance on every rank -> gather on rank 0.  value = all frames / the slowest rank's time."""
    import ctypes

    import torch.distributed as dist

    from underwater_image_enhancement_amd.distributed import comm_device, enhance_sharded, shard_range

    dev = uw.get_device(local)
    H, W, B = args.height, args.width, args.batch
    total = B * world
    cdev = comm_device() if world > 1 else dev.torch_device
    frames = synth_frames(args.dist, total, H, W, dev.torch_device, seed=3000).to(cdev) if rank == 0 else None
    p = dev.params(_lib.SURFACE_SIX, args.strategy)
    lo, hi = shard_range(total, rank, world)
    ws = dev.workspace_for(max(hi - lo, 1), H, W, p)
    out = dev.empty((max(hi - lo, 1), H, W, 3), torch.uint8)

    def compute(x):  # this rank's shard, on the communication device (HBM under RCCL)
        xd = x.to(dev.torch_device)
        n = int(xd.shape[0])
        _lib.check(dev.lib.uwie_enhance_u8(dev._ctx, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, n, H, W,
                                           ctypes.byref(p), ctypes.c_void_p(ws.data_ptr()), ws.numel(), dev.stream()))
        torch.cuda.current_stream().synchronize()
        return out[:n].to(x.device)

    def step():
        if world == 1:
            return compute(frames)
        return enhance_sharded(frames, src=0, compute=compute)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(max(args.warmup, 1)):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    if rank == 0:
        assert res.shape[0] == total
        print(json.dumps({
            "metric": "megapixels/sec enhanced", "value": round(total * H * W * args.steps / elapsed / 1e6, 2),
            "unit": "megapixels/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic",
            "config": {"workload": f"{W}x{H} RGB u8 ({args.dist}), {total} frames held by rank 0, scatter -> strategy{args.strategy} "
                                   f"+ cast correction -> gather ({'gloo rehearsal' if rehearsal else 'RCCL point-to-point'}; "
                                   "BASELINE.json configs[3] layout)",
                       "frames_per_gpu": B, "height": H, "width": W, "parallelism": f"scatter/gather x{world}"}}))
    if world > 1:
        dist.destroy_process_group()


def main_stream(args, uw, _lib, torch, world, rank, local, rehearsal):
    """BASELINE.json configs[4] (SURVEY 8e "for C5 skip the root hop"): every rank owns `batch` frames in PINNED host memory and
    pulls them through its own GPU in chunks -- upload of chunk i+1, enhancement of chunk i and download of chunk i-1 overlap on
    three HIP streams (StreamEnhancer) -- back into pinned host memory.  No rank talks to another on the data path; value = all
    ranks' frames / the slowest rank's time, PCIe included.  --inter f32t selects the reduced-precision transmission
    (uwie_params.inter_dtype = UWIE_INTER_F32T; DESIGN.md section 4 has the stated tolerance and why it is not fp16)."""
    import torch.distributed as dist

    dev = uw.get_device(local)
    H, W, B, chunk = args.height, args.width, args.batch, max(1, min(args.chunk, args.batch))
    over = {"inter_dtype": _lib.INTER_F32T} if args.inter == "f32t" else {}
    host = torch.empty((B, H, W, 3), dtype=torch.uint8, pin_memory=True)
    for b0 in range(0, B, chunk):  # (synthesised on the device chunk by chunk: the frames never have to fit HBM at once)
        n = min(chunk, B - b0)
        host[b0:b0 + n].copy_(synth_frames(args.dist, n, H, W, dev.torch_device, seed=5000 + 100 * rank + b0))
    torch.cuda.synchronize()
    se = uw.StreamEnhancer(H, W, chunk, depth=3, strategy=args.strategy, device=local, **over)
    first = torch.empty((1, H, W, 3), dtype=torch.uint8)

    def step(keep_first=False):
        # results are consumed where they land, in the ring's pinned output buffers (valid until depth - 1 further chunks have
        # been taken): the timed region holds no host-to-host copy -- a single-threaded 200 MB memcpy per chunk was 85 % of the
        # first version's step
        done = 0
        for i, b0 in enumerate(range(0, B, chunk)):
            if len(se._pending) == se.depth - 1:
                r = se.result()
                if done == 0 and keep_first:  # (warm-up only: a 25 MB copy into pageable memory has no place in the timed steps)
                    first.copy_(r[:1])
                done += r.shape[0]
            se.submit_slot(i, src=host[b0:b0 + min(chunk, B - b0)])
        while se._pending:
            r = se.result()
            if done == 0 and keep_first:
                first.copy_(r[:1])
            done += r.shape[0]
        assert done == B

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(max(args.warmup, 1)):
        step(keep_first=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev.torch_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    if rank == 0:
        check = None
        if not args.no_cpu_baseline:  # the first frame against a direct device call on the same bytes (the ring changes nothing)
            direct = uw.enhance(host[:1].numpy(), strategy=args.strategy, device=local, **over)
            check = int((direct.astype(int) - first.numpy().astype(int)).__abs__().max())
        print(json.dumps({
            "metric": "megapixels/sec enhanced", "value": round(world * B * H * W * args.steps / elapsed / 1e6, 2),
            "unit": "megapixels/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32/f32" if args.inter == "f32t" else "f32/f64", "data": "synthetic",
            "config": {"workload": f"{W}x{H} RGB u8 ({args.dist}) stream: {B} frames per GPU from pinned host memory in chunks of "
                                   f"{chunk}, upload / strategy{args.strategy} + cast correction / download overlapped on three streams, "
                                   f"results left in the ring's pinned host buffers (PCIe-inclusive; BASELINE.json configs[4]; intermediates "
                                   f"{args.inter}{'; gloo rehearsal, every rank on GPU 0' if rehearsal else ''})",
                       "frames_per_gpu": B, "chunk": chunk, "height": H, "width": W, "parallelism": f"independent streams x{world}"},
            "stream_vs_direct_max_lsb": check}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--strategy", type=int, default=2)
    ap.add_argument("--dist", choices=("underwater", "uniform", "hazy"), default="underwater",
                    help="synthetic input distribution (SURVEY.md section 8d); the headline number uses underwater")
    ap.add_argument("--mode", choices=("local", "scatter", "stream"), default="local",
                    help="local: every rank enhances frames it already holds (headline, weak scaling).  scatter: rank 0 holds the "
                         "whole batch in HBM, each step = scatter over RCCL -> enhance -> gather (BASELINE.json configs[3]).  stream: "
                         "every rank streams its own frames from pinned host memory through its GPU and back (configs[4])")
    ap.add_argument("--chunk", type=int, default=8, help="--mode stream: frames per upload / enhance / download chunk")
    ap.add_argument("--inter", choices=("f64", "f32t"), default="f64",
                    help="--mode stream: number format of the refined transmission (uwie_params.inter_dtype)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel times of the recorded warm-up step to stderr")
    ap.add_argument("--no-extras", action="store_true", help="skip the other input distributions and the 1080p batch=1 case")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under torchrun: start one fresh process per GPU ourselves, BEFORE anything here touches the GPU (a process
        # that has initialised HIP must not fork/exec workers), and hand back rank 0's line and the worst exit code.
        sys.exit(spawn_ranks(args.gpus))

    import torch

    import underwater_image_enhancement_amd as uw
    from underwater_image_enhancement_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # UWIE_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo for the timing reduction -- exercises the multi-rank control
    # flow on a one-GPU box (RCCL refuses two ranks on one device); not a measurement.
    rehearsal = os.environ.get("UWIE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node "
                 f"{args.gpus} (or without a launcher: bench.py starts the ranks itself)")
    if args.mode == "scatter":
        return main_scatter(args, uw, _lib, torch, world, rank, local, rehearsal)
    if args.mode == "stream":
        return main_stream(args, uw, _lib, torch, world, rank, local, rehearsal)

    dev = uw.get_device(local)
    H, W, B = args.height, args.width, args.batch
    frames = synth_frames(args.dist, B, H, W, dev.torch_device, seed=1000 * 2 + rank)
    p = dev.params(_lib.SURFACE_SIX, args.strategy)
    ws = dev.workspace_for(B, H, W, p)
    out = dev.empty((B, H, W, 3), torch.uint8)

    import ctypes

    def step():
        _lib.check(dev.lib.uwie_enhance_u8(dev._ctx, ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           None, B, H, W, ctypes.byref(p), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                           dev.stream()))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Warm-up (untimed).  Its last step is recorded kernel by kernel to find the dominant kernel: the timed region then
    # carries HIP events around that kernel only (an event pair per launch costs ~9 us of stream time: 1.4 ms of a
    # 22 ms step when all 150 launches are recorded, which would be charged to `value`).
    for _ in range(max(args.warmup - 1, 1)):  # at least one plain step before the recorded one (first-touch costs)
        step()
    torch.cuda.synchronize()
    dev.profile(True)
    step()
    table = dev.profile_rows()  # {kernel: (ms, launches)} of one step, every kernel
    dev.profile(False)
    name = max(table.items(), key=lambda kv: kv[1][0])[0]
    barrier()
    dev.profile(True, only=name)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rows = dev.profile_rows()
    dev.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev.torch_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    if rank == 0:
        px_step = B * H * W
        value = world * px_step * args.steps / elapsed / 1e6
        ms, calls = rows[name]  # the dominant kernel, measured inside the timed region
        per_launch_ms = ms / calls
        launches_per_step = calls / args.steps
        cover = 1.0
        if name in ("k_guided_split", "k_guided_pipe"):  # the two launches of the guided filter share the rows of a frame
            import ctypes

            r0, rows = ctypes.c_int(0), ctypes.c_int(0)
            _lib.check(dev.lib.uwie_guided_plan(B, H, W, int(p.gf_ksize), ctypes.byref(r0), ctypes.byref(rows)))
            cover = rows.value / H if name == "k_guided_split" else 1.0 - rows.value / H
        bytes_launch = KERNEL_BYTES_PER_PX.get(name, 0) * cover * px_step / launches_per_step
        achieved = bytes_launch / (per_launch_ms * 1e-3) / 1e9
        kernel_ms = sum(v[0] for v in table.values())  # all kernels, from the recorded warm-up step
        if args.kernel_table:
            for kname, (kms, kcalls) in sorted(table.items(), key=lambda kv: -kv[1][0]):
                print(f"  {kms:8.3f} ms  {kcalls:4d} launches  {kname}", file=sys.stderr)
        result = {
            "metric": "megapixels/sec enhanced", "value": round(value, 2), "unit": "megapixels/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} RGB u8 ({args.dist}), batch={B} per GPU, strategy{args.strategy} + cast "
                                   "correction (canonical enhance, BASELINE.json configs[2])",
                       "frames_per_gpu": B, "height": H, "width": W, "parallelism": f"batch-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(name, H, W, B, args.strategy, launches_per_step, args.dist),
                         "traffic_source": f"from committed profile {TRAFFIC_PROFILE} (rocprofv3 PMC passes of this command; not "
                                           "measured in this run)", "kernel": name,
                         "kernel_ms_per_launch": round(per_launch_ms, 4), "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": bytes_launch, "rows_covered_frac": round(cover, 4),
                         "kernel_share_of_step": round(ms / args.steps / (elapsed / args.steps * 1e3), 4),
                         "pipeline_achieved": round(PIPELINE_BYTES_PER_PX * px_step * args.steps / elapsed / 1e9, 2),
                         "pipeline_frac": round(PIPELINE_BYTES_PER_PX * px_step * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                         "sum_kernel_ms_per_step": round(kernel_ms, 3),
                         "note": "kernel_ms_per_launch: HIP events around this kernel inside the timed region; "
                                 "sum_kernel_ms_per_step: all kernels, recorded on the last warm-up step"},
        }
        if world == 1 and not args.no_cpu_baseline:
            n_cmp = min(B, 8)
            result["cpu_baseline"] = cpu_baseline(frames[:n_cmp].cpu().numpy(), out[:n_cmp].cpu().numpy(), args.strategy, args.dist)
        if world == 1 and not args.no_extras:
            result["extras"] = extras(dev, args, torch, _lib)
        if os.environ.get("UWIE_BENCH_KERNELS"):
            top = sorted(table.items(), key=lambda kv: -kv[1][0])
            print("# per-kernel ms/step (recorded warm-up step): " + ", ".join(f"{k}={v[0]:.3f}({v[1]})" for k, v in top),
                  file=sys.stderr)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
