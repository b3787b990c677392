"""Batch sharding across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Frames are independent -- every global statistic of the pipeline (cast means, atmospheric light, percentiles, CLAHE
tile histograms) is per image -- so the path shards by frames with NO collective on the data path.  Two layouts:

* each rank already holds its frames (the bench, a video pipeline with one decoder per GPU): call ``enhance`` locally;
* one rank holds the whole batch (BASELINE.json configs[3]): ``scatter_frames`` -> local enhance -> ``gather_frames``.
  Root sends each peer its contiguous uint8 shard with ONE group of point-to-point sends (``batch_isend_irecv`` =
  ncclGroupStart / ncclSend x 7 / ncclGroupEnd: 7 peers = 7 distinct xGMI links in parallel, per-link bound), peers
  send results back the same way; nothing is reduced.

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


def comm_device(backend: str | None = None) -> torch.device:
    """The device every tensor handed to a collective must live on: the rank's current ROCm device under "nccl"
    (= RCCL, which rejects CPU tensors), the CPU under "gloo".  The same on every rank, whoever holds the data."""
    backend = backend or dist.get_backend()
    if str(backend).lower() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _meta(frames, src, device):
    """Broadcast (B, H, W) from the source rank."""
    shape = torch.zeros(3, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        shape[:] = torch.tensor(frames.shape[:3], dtype=torch.int64)
    dist.broadcast(shape, src=src)
    return tuple(int(v) for v in shape)


def _p2p(ops):
    """Issue a list of ``dist.P2POp`` as ONE group (``batch_isend_irecv``: ncclGroupStart ... ncclSend/Recv x N ...
    ncclGroupEnd under RCCL) and wait for all of them."""
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def scatter_frames(frames, src: int = 0, device=None):
    """Rank ``src`` holds uint8 ``[B,H,W,3]`` (host or device memory); every rank returns its shard (possibly 0
    frames).  Everything that travels lives on the backend's device (``comm_device()``: HBM under RCCL, host memory
    under gloo) on every rank; ``device`` only says where the returned local shard is finally placed (default: the
    communication device) -- the copy happens after the receive, never before the send.  A host batch on the root is
    moved to the communication device shard by shard, so the root never holds a second full copy.  The root's sends go
    out as one group (7 peers = 7 xGMI links in parallel)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = comm_device()
    B, H, W = _meta(frames, src, comm)
    lo, hi = shard_range(B, rank, world)
    if rank == src:
        ops, keep = [], []
        for peer in range(world):
            if peer == src:
                continue
            plo, phi = shard_range(B, peer, world)
            if phi > plo:
                shard = frames[plo:phi].to(comm, non_blocking=True).contiguous()
                keep.append(shard)  # alive until the send has completed
                ops.append(dist.P2POp(dist.isend, shard, peer))
        local = frames[lo:hi].to(comm).contiguous()
        _p2p(ops)
    else:
        local = torch.empty((hi - lo, H, W, 3), dtype=torch.uint8, device=comm)
        if hi > lo:
            _p2p([dist.P2POp(dist.irecv, local, src)])
    return local if device is None else local.to(torch.device(device))


def gather_frames(local, n_frames: int, dst: int = 0):
    """Inverse of scatter_frames: rank ``dst`` returns ``[n_frames,H,W,3]`` (on the communication device), the
    others ``None``.  The root's receives are one group."""
    rank, world = dist.get_rank(), dist.get_world_size()
    local = local.to(comm_device())
    # Ranks with an empty shard skip the point-to-point group below.  Under NCCL / RCCL a batched P2P that is the FIRST
    # communication on a process group must involve every rank (it creates the communicator); scatter_frames and
    # enhance_sharded broadcast first, a stand-alone gather does not -- so it opens with a collective of its own (4 bytes; it
    # also tells every rank the root's frame shape is not needed from it: nothing else is exchanged here).
    dist.broadcast(torch.zeros(1, dtype=torch.int32, device=local.device), src=dst)
    if rank != dst:
        if local.shape[0] > 0:
            _p2p([dist.P2POp(dist.isend, local.contiguous(), dst)])
        return None
    H, W = int(local.shape[1]), int(local.shape[2])
    out = torch.empty((n_frames, H, W, 3), dtype=local.dtype, device=local.device)
    lo, hi = shard_range(n_frames, rank, world)
    out[lo:hi] = local
    ops = []
    for peer in range(world):
        if peer == dst:
            continue
        plo, phi = shard_range(n_frames, peer, world)
        if phi > plo:
            ops.append(dist.P2POp(dist.irecv, out[plo:phi], peer))
    _p2p(ops)
    return out


def enhance_sharded(frames, src: int = 0, compute=None, device=None, **kwargs):
    """Root-held batch -> scatter -> per-rank enhance -> gather back to the root (BASELINE.json configs[3]).  Returns the
    enhanced batch on ``src`` (on the communication device) and ``None`` elsewhere.  ``src`` is the rank that holds
    ``frames`` (the other ranks pass ``None``); ``device`` is where the local shard is handed to ``compute`` (default: the
    communication device; the collectives themselves always run on ``comm_device()``); ``kwargs`` go to ``api.enhance``
    (strategy, cast_correct, overrides)."""
    if compute is None:
        from .api import enhance

        def compute(x):
            return enhance(x, **kwargs)

    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=comm_device())  # (a tensor broadcast: no pickling, no device guessing)
    if rank == src:
        n[0] = int(frames.shape[0])
    dist.broadcast(n, src=src)
    local = scatter_frames(frames if rank == src else None, src=src, device=device)
    out = compute(local) if local.shape[0] > 0 else local
    if not torch.is_tensor(out):
        out = torch.as_tensor(out)
    return gather_frames(out, int(n[0]), dst=src)
