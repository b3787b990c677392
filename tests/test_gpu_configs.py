"""Full-size coverage of BASELINE.json's configurations on ONE MI355X (round 3): the launch shapes that were only timed
before are compared with the oracle or with the single-frame calls here.

* configs[3]'s per-GPU share: 1080p x 128 in one call;
* the RCCL path executed once: a world-size-1 "nccl" process group inside the pytest process, `enhance_sharded` with its
  default compute (the HIP path on a ROCm tensor);
* one 1080p frame through strategies 1, 3, 4, 5, 6 and the five dict strategies, strategies 1 and 3 on one 4K frame,
  against the oracle (launch-shape dependent kernels: guided filter for k = 20 / 10 with many bands, strategy 3's
  collecting sweep on stored planes, code-domain CLAHE with 480 x 270 tiles, the float64 selection);
* configs[4]'s host streaming at 4K, chunk 8;
* a frame above 22 MP (the quadtree's chunk-sum staging must not size its LDS by the frame).
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uw():
    import underwater_image_enhancement_amd as uw

    return uw


@pytest.fixture(scope="module")
def orc():
    from oracle import uwie_oracle

    return uwie_oracle


def underwater(rng, H, W, gains, noise=0.02):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    ph = rng.random(3) * 6.283
    field = 0.55 + 0.25 * (np.sin(xx / (W / 9.0) + ph[0]) * np.cos(yy / (H / 7.0) + ph[1])
                           + 0.5 * np.sin((xx + 2 * yy) / (W / 5.0) + ph[2])) / 1.5
    f = field[:, :, None] * np.array(gains, np.float32)[None, None, :] + rng.normal(0, noise, (H, W, 3)).astype(np.float32)
    return np.clip(np.floor(255 * f), 0, 255).astype(np.uint8)


def device_frames(dev, B, H, W, seed):
    """B synthetic underwater frames built on the device (greenish / bluish alternating)."""
    import torch

    g = torch.Generator(device=dev.torch_device).manual_seed(seed)
    yy = torch.arange(H, device=dev.torch_device, dtype=torch.float32)[:, None]
    xx = torch.arange(W, device=dev.torch_device, dtype=torch.float32)[None, :]
    frames = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev.torch_device)
    for b in range(B):
        ph = torch.rand(4, generator=g, device=dev.torch_device) * 6.283
        field = 0.55 + 0.125 * (torch.sin(xx / (W / 9.0) + ph[0]) * torch.cos(yy / (H / 7.0) + ph[1])
                                + 0.5 * torch.sin((xx + 2 * yy) / (W / 5.0) + ph[2]))
        gains = (0.45, 0.85, 0.80) if b % 2 == 0 else (0.45, 0.75, 0.90)
        for c in range(3):
            ch = field * gains[c] + torch.randn((H, W), generator=g, device=dev.torch_device) * 0.02
            frames[b, :, :, c] = torch.clamp(torch.floor(ch * 255.0), 0, 255).to(torch.uint8)
    return frames


def assert_bytes(got, want, what, max_diff_bytes=0):
    assert got.shape == want.shape and got.dtype == np.uint8
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"{what}: max |delta| = {d.max()} LSB"  # BASELINE.json: <= 1 LSB
    n = int(np.count_nonzero(d))
    assert n <= max_diff_bytes, f"{what}: {n} bytes differ by 1 LSB"


# ------------------------------------------------------------------ configs[3]: the per-GPU share
def test_config3_share_1080p_batch128(uw, orc):
    """1024 1080p frames over 8 GPUs = 128 frames per GPU in one call (11.4 GB of workspace, sized by
    the library).  The batch equals the single-frame calls on sampled frames (first, last, a greenish, a bluish, a noise
    frame) and one frame equals the oracle."""
    import torch

    dev = uw.get_device()
    B, H, W = 128, 1080, 1920
    frames = device_frames(dev, B, H, W, 3128)
    g = torch.Generator(device=dev.torch_device).manual_seed(5)
    frames[77] = torch.randint(0, 256, (H, W, 3), generator=g, device=dev.torch_device, dtype=torch.uint8)
    out = uw.enhance(frames)
    assert out.shape == frames.shape and out.dtype == torch.uint8
    for b in (0, 1, 64, 77, 126, 127):
        assert torch.equal(out[b], uw.enhance(frames[b:b + 1])[0]), f"frame {b} of the batch differs from its single call"
    assert_bytes(out[3].cpu().numpy(), orc.enhance_u8(frames[3].cpu().numpy(), 2), "frame 3 of the 1080p x 128 batch")
    del frames, out
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ RCCL executed once
def test_rccl_world1_enhance_sharded_default_compute(uw):
    """north_star: "RCCL scatter/gather over xGMI".  One GPU cannot exchange anything, but a world-size-1 "nccl" group runs
    every RCCL call of the path that does not need a peer: communicator creation, the device-tensor broadcasts of
    `_meta` / `enhance_sharded`, `comm_device()`, the grouped point-to-point helpers with empty peer lists, and the default
    compute (`api.enhance` on a ROCm tensor).  The result must equal the direct call."""
    import torch
    import torch.distributed as dist

    from underwater_image_enhancement_amd.distributed import comm_device, enhance_sharded, gather_frames, scatter_frames

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    assert not dist.is_initialized()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        assert comm_device() == torch.device("cuda", 0)
        rng = np.random.default_rng(31)
        host = np.stack([underwater(rng, 270, 480, (0.45, 0.85, 0.80)), underwater(rng, 270, 480, (0.45, 0.75, 0.90)),
                         rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)])
        want = uw.enhance(host, strategy=2)
        # a device-resident batch and a host-resident batch (moved to HBM shard by shard)
        for frames in (torch.from_numpy(host).cuda(), torch.from_numpy(host)):
            out = enhance_sharded(frames, src=0, strategy=2)
            assert out.is_cuda and out.dtype == torch.uint8 and np.array_equal(out.cpu().numpy(), want)
        # an all-reduce on the device proves the communicator is RCCL's and alive
        one = torch.ones(4, device="cuda")
        dist.all_reduce(one)
        assert torch.equal(one.cpu(), torch.ones(4))
        local = scatter_frames(torch.from_numpy(host), src=0)
        assert local.is_cuda and torch.equal(gather_frames(local, 3, dst=0).cpu(), torch.from_numpy(host))
        assert scatter_frames(torch.from_numpy(host), src=0, device="cpu").device.type == "cpu"
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ full-size strategies against the oracle
@pytest.mark.parametrize("strategy", [1, 3, 4, 5, 6])
def test_1080p_six_strategies_match_oracle(uw, orc, strategy):
    """configs[1]'s frame size through the strategies the full-size tests did not cover (strategy 2 is in
    test_gpu_enhance.py): k = 20 / 10 guided filter over many bands, strategy 3's collecting sweep on stored planes,
    the code-domain strategies with 240 x 135 CLAHE tiles."""
    u8 = underwater(np.random.default_rng(1080 + strategy), 1080, 1920, (0.45, 0.85, 0.80) if strategy % 2 else (0.45, 0.75, 0.90))
    # strategies 1 end in x**gamma per LUT entry and the guided filter's 1e-11 on t: <= 1 LSB, practically identical
    assert_bytes(uw.enhance(u8, strategy=strategy), orc.enhance_u8(u8, strategy), f"1080p strategy {strategy}", max_diff_bytes=16)


@pytest.mark.parametrize("name", ["strong_dehazing", "medium_dehazing", "light_enhancement", "clahe_enhancement",
                                  "histogram_equalization"])
def test_1080p_dict_strategies_match_oracle(uw, orc, name):
    """enhancement_strategies.py's five strategies with Config.STRATEGIES' parameters (config.py:28-75) on a 1080p frame:
    `(enhanced * 255).astype(uint8)` as main.py:155 computes it."""
    u8 = underwater(np.random.default_rng(2080 + len(name)), 1080, 1920, (0.45, 0.85, 0.80))
    x = orc.normalise_u8(u8)
    params = orc.CONFIG_STRATEGIES[name]
    want = orc.DictStrategyOracle.run(x, name, params)
    got = uw.EnhancementStrategies.apply_strategy(x, name, params)
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.abs(got - want).max() < 1e-9
    assert_bytes((got * 255).astype(np.uint8), (want * 255).astype(np.uint8), f"1080p dict {name}",
                 max_diff_bytes=16 if params.get("apply_gamma", False) else 0)


@pytest.mark.parametrize("strategy", [1, 3])
def test_4k_strategies_1_and_3_match_oracle(uw, orc, strategy):
    """configs[2]'s frame size through the other two dehazing strategies: the even-width guided filter (k = 20, k = 10) with
    the 4K band plan, strategy 3's stored planes and four-percentile selection."""
    u8 = underwater(np.random.default_rng(4000 + strategy), 2160, 3840, (0.45, 0.75, 0.90))
    assert_bytes(uw.enhance(u8, strategy=strategy), orc.enhance_u8(u8, strategy), f"4K strategy {strategy}", max_diff_bytes=32)


# ------------------------------------------------------------------ configs[4]: host streaming at 4K
def test_stream_enhancer_4k_chunk8(uw):
    """4K frames in host memory through the pinned three-stream ring in chunks of 8 (two full chunks and a short one, more
    chunks than... ring slots are reused from the fourth on): equal to direct calls."""
    import torch

    dev = uw.get_device()
    H, W = 2160, 3840
    frames = device_frames(dev, 20, H, W, 44).cpu()
    se = uw.StreamEnhancer(H, W, chunk=8, depth=2)
    outs = [o.clone() for o in se.run(frames[i:i + 8] for i in range(0, 20, 8))]
    assert [o.shape[0] for o in outs] == [8, 8, 4]
    got = torch.cat(outs)
    for lo in (0, 8, 16):
        want = uw.enhance(frames[lo:lo + 4].to(dev.torch_device))
        assert torch.equal(got[lo:lo + 4], want.cpu()), f"streamed frames {lo}..{lo + 3} differ from the direct call"
    del se, frames
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ frames above 22 MP
def test_24mp_still_runs_and_matches_the_oracle_quadtree(uw, orc):
    """A 6000 x 4000 still has 733 NumPy buffers per level-0 quadrant: the chunk-sum staging of the quadtree kernels goes
    through fixed-size LDS pieces (round 2 sized a launch's LDS by the frame and failed above 22.3 MP).  The atmospheric
    light (every level's block and the brightest pixel) must be the oracle's, and the whole pipeline must run."""
    H, W = 4000, 6000
    u8 = underwater(np.random.default_rng(24), H, W, (0.45, 0.85, 0.80))
    dev = uw.get_device()
    want_A = orc.atmospheric_light(orc.normalise_u8(u8))  # no cast correction: the quadtree on the plain frame
    got_A = dev.atmospheric_light(dev.tensor(u8[None]), None).cpu().numpy()[0]
    assert np.array_equal(got_A, np.asarray(want_A, np.float32).reshape(-1)[:3])
    out = uw.enhance(u8)
    assert out.shape == u8.shape and out.dtype == np.uint8
    # size-independent property: the frame inside a batch of two equals the single call
    pair = np.stack([u8, u8[::-1].copy()])
    assert np.array_equal(uw.enhance(pair)[0], out)


def test_bench_stream_mode_two_ranks_rehearsal():
    """BASELINE.json configs[4] control flow at N > 1: `bench.py --mode stream --gpus 2` starts two ranks
    itself; each streams its own pinned shard through its own StreamEnhancer (three HIP streams) and the elapsed times meet in a
    MAX reduction.  One-GPU box: UWIE_BENCH_REHEARSAL puts both ranks on GPU 0 with gloo for the reduction -- a rehearsal of the
    control flow, not a measurement (RCCL refuses two ranks on one device).  The line must carry both ranks' frames and a stream
    result identical to a direct call; --inter f32t runs the reduced-precision transmission through the same path."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UWIE_BENCH_REHEARSAL="1")
    for inter in ("f64", "f32t"):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--mode", "stream", "--gpus", "2", "--height", "256", "--width", "320",
               "--batch", "6", "--chunk", "4", "--steps", "1", "--warmup", "1", "--inter", inter]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, res.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["frames_per_gpu"] == 6 and d["config"]["chunk"] == 4
        assert d["stream_vs_direct_max_lsb"] == 0 and d["value"] > 0 and "configs[4]" in d["config"]["workload"]


@pytest.mark.parametrize("mode", ["local", "scatter"])
def test_bench_two_ranks_under_the_drivers_launcher_rehearsal(mode):
    """The driver's N > 1 launch line, word for word (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`), for the headline mode and for configs[3]'s scatter /
    gather mode: RANK / LOCAL_RANK / WORLD_SIZE come from the launcher, every rank runs the HIP path on its frames, the elapsed
    times meet in a MAX reduction and rank 0 alone prints the line, whose `value` counts both ranks' frames.  One-GPU box:
    UWIE_BENCH_REHEARSAL puts both ranks on GPU 0 with gloo for the reduction and the transfers (RCCL refuses two ranks on one
    device) -- a rehearsal of the control flow, not a measurement."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, UWIE_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    H, W, B, K = 256, 320, 6, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", "1",
           "--mode", mode, "--height", str(H), "--width", str(W), "--batch", str(B)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == K and d["scaling"] == "weak" and d["value"] > 0
    frames_per_step = d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / (H * W)
    assert abs(frames_per_step - 2 * B) < 0.01 * 2 * B, (frames_per_step, d)  # both ranks' frames (scatter: the root's batch)
    assert "cpu_baseline" not in d  # the CPU oracle is timed by rank 0 at N = 1 only
    if mode == "local":
        assert d["roofline"]["kernel"] and d["config"]["parallelism"] == "batch-shard x2"
    else:
        assert d["config"]["parallelism"] == "scatter/gather x2" and "gloo rehearsal" in d["config"]["workload"]
