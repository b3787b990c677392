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

from underwater_image_enhancement_amd.distributed import enhance_sharded, gather_frames, scatter_frames, shard_range


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
        local = scatter_frames(full if rank == 0 else None, src=0)
        lo, hi = shard_range(n_frames, rank, world)
        assert local.shape[0] == hi - lo and torch.equal(local, full[lo:hi])
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
