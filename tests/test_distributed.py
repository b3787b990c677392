"""World-size-2 checks of the batch sharding (gloo on CPU): partition, scatter, per-rank compute, gather.

The per-rank compute is injected (a frame-wise NumPy function standing in for the HIP path, which needs a GPU);
what is under test is that every frame is processed exactly once, by one rank, and lands back in order.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from underwater_image_enhancement_amd.distributed import comm_device, enhance_sharded, gather_frames, scatter_frames, shard_range


def test_shard_range_partitions_every_batch():
    for n in (0, 1, 2, 5, 64, 1024, 1023):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(1024, 3, 8) == (384, 512)  # BASELINE.json configs[3]: 128 frames per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_comm_device_follows_the_backend(monkeypatch):
    """RCCL ("nccl") only moves device tensors: every rank -- also the ones that hold no frames -- must put its shape
    broadcast, its receive buffer and its shards on its own ROCm device; gloo stays on the CPU."""
    assert comm_device("gloo") == torch.device("cpu")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert comm_device("nccl") == torch.device("cuda", 3)
    assert comm_device("NCCL") == torch.device("cuda", 3)


def _frame_op(batch):
    """Stand-in per-frame computation with a per-image global statistic (like the real pipeline)."""
    x = batch.numpy().astype(np.int32)
    out = np.empty_like(x)
    for i in range(x.shape[0]):
        out[i] = (x[i] * 3 + int(x[i].max())) % 251
    return torch.from_numpy(out.astype(np.uint8))


def _worker(rank, world, port, n_frames, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)
        full = torch.from_numpy(rng.integers(0, 256, (n_frames, 12, 10, 3), dtype=np.uint8))
        assert comm_device() == torch.device("cpu")
        local = scatter_frames(full if rank == 0 else None, src=0)  # default device: the backend's
        lo, hi = shard_range(n_frames, rank, world)
        assert local.device.type == "cpu" and local.shape[0] == hi - lo and torch.equal(local, full[lo:hi])
        back = gather_frames(local, n_frames, dst=0)
        out = enhance_sharded(full if rank == 0 else None, src=0, compute=_frame_op)
        if rank == 0:
            assert torch.equal(back, full)
            assert torch.equal(out, _frame_op(full))
            with open(result_path, "w") as f:
                f.write("ok")
        else:
            assert back is None and out is None
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_frames", [5, 1, 8])
def test_scatter_compute_gather_world2(tmp_path, n_frames):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), n_frames, str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"


def _gather_first_worker(rank, world, port, n_frames, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_frames, rank, world)
        local = torch.full((hi - lo, 6, 5, 3), 10 + rank, dtype=torch.uint8)
        back = gather_frames(local, n_frames, dst=0)  # the FIRST communication on the group; some shards are empty
        if rank == 0:
            want = torch.cat([torch.full((shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0], 6, 5, 3), 10 + r,
                                         dtype=torch.uint8) for r in range(world)])
            assert torch.equal(back, want)
            with open(result_path, "w") as f:
                f.write("ok")
        else:
            assert back is None
    finally:
        dist.destroy_process_group()


def test_gather_as_the_first_communication_with_empty_shards(tmp_path):
    """gather_frames' grouped point-to-point skips ranks whose shard is empty; as the first communication on a
    group that is undefined under NCCL / RCCL, so it opens with a collective of its own.  World 3, two frames: rank 2 has none."""
    result = tmp_path / "result.txt"
    mp.spawn(_gather_first_worker, args=(3, _free_port(), 2, str(result)), nprocs=3, join=True)
    assert result.read_text() == "ok"


# ------------------------------------------------------------------ bench.py launch behaviour (no GPU needed)
def _load_bench():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("uwie_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_spawns_one_fresh_process_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher must produce N ranks itself (never a silent N=1 run): N children
    with the torchrun environment, rendezvous on 127.0.0.1, started before the parent has touched the GPU."""
    import subprocess
    import sys

    bench = _load_bench()
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, env))

        def wait(self):
            return 0 if int(started[0][1]["WORLD_SIZE"]) == 4 else 3

    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None: FakeProc(cmd, env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0 and len(started) == 4
    assert not torch.cuda.is_initialized()
    for r, (cmd, env) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "2"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "4", "127.0.0.1")
    assert len({env["MASTER_PORT"] for _, env in started}) == 1
    # a failing rank fails the run
    started.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 3 and len(started) == 2


def test_bench_refuses_a_world_size_that_differs_from_gpus(monkeypatch):
    import sys

    bench = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code not in (0, None) and "WORLD_SIZE=1" in str(exc.value.code)
