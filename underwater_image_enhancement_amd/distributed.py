"""Batch sharding across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Frames are independent -- every global statistic of the pipeline (cast means, atmospheric light, percentiles, CLAHE
tile histograms) is per image -- so the path shards by frames with NO collective on the data path.  Two layouts:

* each rank already holds its frames (the bench, a video pipeline with one decoder per GPU): call ``enhance`` locally;
* one rank holds the whole batch (BASELINE.json configs[3]): ``scatter_frames`` -> local enhance -> ``gather_frames``.
  Root sends each peer its contiguous uint8 shard with point-to-point sends (7 peers = 7 distinct xGMI links in
  parallel, per-link bound), peers send results back the same way; nothing is reduced.

``compute`` is injectable so the CPU (gloo) tests can exercise the sharding logic without a GPU; the product default
is the HIP path (``api.enhance``), which raises without a device -- there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split of the batch dimension: the first ``n % world`` ranks get one extra frame."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _meta(frames, src, device):
    """Broadcast (B, H, W) from the source rank."""
    shape = torch.zeros(3, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        shape[:] = torch.tensor(frames.shape[:3], dtype=torch.int64)
    dist.broadcast(shape, src=src)
    return tuple(int(v) for v in shape)


def scatter_frames(frames, src: int = 0, device=None):
    """Rank ``src`` holds uint8 ``[B,H,W,3]``; every rank returns its shard (possibly 0 frames) on ``device``."""
    rank, world = dist.get_rank(), dist.get_world_size()
    device = device or (frames.device if frames is not None else torch.device("cpu"))
    B, H, W = _meta(frames, src, device)
    lo, hi = shard_range(B, rank, world)
    if rank == src:
        reqs = []
        for peer in range(world):
            if peer == src:
                continue
            plo, phi = shard_range(B, peer, world)
            if phi > plo:
                reqs.append(dist.isend(frames[plo:phi].contiguous(), dst=peer))
        local = frames[lo:hi].contiguous()
        for r in reqs:
            r.wait()
        return local
    local = torch.empty((hi - lo, H, W, 3), dtype=torch.uint8, device=device)
    if hi > lo:
        dist.recv(local, src=src)
    return local


def gather_frames(local, n_frames: int, dst: int = 0):
    """Inverse of scatter_frames: rank ``dst`` returns ``[n_frames,H,W,3]``, the others ``None``."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != dst:
        if local.shape[0] > 0:
            dist.send(local.contiguous(), dst=dst)
        return None
    H, W = int(local.shape[1]), int(local.shape[2])
    out = torch.empty((n_frames, H, W, 3), dtype=local.dtype, device=local.device)
    lo, hi = shard_range(n_frames, rank, world)
    out[lo:hi] = local
    reqs = []
    for peer in range(world):
        if peer == dst:
            continue
        plo, phi = shard_range(n_frames, peer, world)
        if phi > plo:
            reqs.append(dist.irecv(out[plo:phi], src=peer))
    for r in reqs:
        r.wait()
    return out


def enhance_sharded(frames, src: int = 0, compute=None, device=None, **kwargs):
    """Root-held batch -> scatter -> per-rank enhance -> gather back to the root.  Returns the enhanced batch on
    ``src`` and ``None`` elsewhere.  ``kwargs`` go to ``api.enhance`` (strategy, cast_correct, overrides)."""
    if compute is None:
        from .api import enhance

        def compute(x):
            return enhance(x, **kwargs)

    rank = dist.get_rank()
    n = [int(frames.shape[0]) if rank == src else 0]
    dist.broadcast_object_list(n, src=src)
    local = scatter_frames(frames if rank == src else None, src=src, device=device)
    out = compute(local) if local.shape[0] > 0 else local
    return gather_frames(out, n[0], dst=src)
