"""Host-memory streaming (BASELINE.json configs[4]): frames that live in host memory go through
the device in chunks, with the host-to-device copy of chunk i+1 and the device-to-host copy of chunk i-1 overlapped
with the enhancement of chunk i.

Three HIP streams (upload, compute, download) and a ring of ``depth`` slots; each slot owns a pinned input buffer, a
pinned output buffer and their device twins.  Ordering is by events only; the host blocks when it needs a slot back
(its previous download has to be finished) and when it hands a result to the caller.  PyTorch supplies the pinned
buffers, streams and events: the enhancement itself is the C ABI call ``uwie_enhance_u8``.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check
from .runtime import _ptr, get_device


class StreamEnhancer:
    """``for out in StreamEnhancer(H, W, chunk).run(chunks)``: ``chunks`` yields uint8 ``[n<=chunk, H, W, 3]`` host
    arrays (NumPy or CPU tensors), ``out`` is the enhanced chunk as a uint8 CPU tensor view of a pinned buffer that
    stays valid until ``depth - 1`` further chunks have been taken.  A producer that can write straight into pinned
    memory uses ``input_slot(i)`` / ``submit_slot(i, n)`` / ``result()`` and saves the staging copy."""

    def __init__(self, H: int, W: int, chunk: int, depth: int = 3, strategy: int = 2, cast_correct: bool = True,
                 device: int | None = None, **overrides):
        if depth < 2:
            raise ValueError("depth must be >= 2 to overlap anything")
        self.dev = get_device(device)
        self.H, self.W, self.chunk, self.depth = int(H), int(W), int(chunk), int(depth)
        d = self.dev
        self.p = d.params(_lib.SURFACE_SIX, int(strategy), cast_correct=int(bool(cast_correct)), **overrides)
        shape = (self.chunk, self.H, self.W, 3)
        self.h_in = [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
        self.h_out = [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
        self.d_in = [torch.empty(shape, dtype=torch.uint8, device=d.torch_device) for _ in range(depth)]
        self.d_out = [torch.empty(shape, dtype=torch.uint8, device=d.torch_device) for _ in range(depth)]
        nbytes = d.lib.uwie_workspace_bytes_ctx(d._ctx, self.chunk, self.H, self.W, ctypes.byref(self.p))
        self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=d.torch_device)  # one: compute is serial anyway
        self.s_up, self.s_run, self.s_down = (torch.cuda.Stream(d.torch_device) for _ in range(3))
        self.e_up = [torch.cuda.Event() for _ in range(depth)]
        self.e_run = [torch.cuda.Event() for _ in range(depth)]
        self.e_down = [torch.cuda.Event() for _ in range(depth)]
        self._used = [False] * depth
        self._pending = []            # (slot, n) in submission order
        self._next = 0

    # ------------------------------------------------------------------ zero-copy producer interface
    def input_slot(self, i: int):
        """Pinned input buffer of slot ``i % depth``; blocks until the slot's previous result has been downloaded."""
        slot = i % self.depth
        if any(s == slot for s, _ in self._pending):
            raise RuntimeError("slot still holds an untaken result: call result() first")
        self.e_down[slot].synchronize()
        return self.h_in[slot]

    def submit_slot(self, i: int, n: int | None = None, src=None):
        """Start chunk ``i``: upload, enhance, download.  ``src`` (optional): a PINNED uint8 host tensor ``[n, H, W, 3]`` to upload
        from instead of the slot's own input buffer -- frames that already sit in pinned memory (a capture ring, a decoder's
        output) skip the staging copy altogether.  The caller keeps ``src`` untouched until the chunk's result is taken."""
        slot = i % self.depth
        if src is not None:
            if not src.is_pinned() or src.dtype != torch.uint8 or tuple(src.shape[1:]) != (self.H, self.W, 3) or src.shape[0] > self.chunk:
                raise ValueError("src must be a pinned uint8 [<= chunk, H, W, 3] host tensor")
            n = int(src.shape[0])
            if any(s == slot for s, _ in self._pending):
                raise RuntimeError("slot still holds an untaken result: call result() first")
            self.e_down[slot].synchronize()
        n = self.chunk if n is None else int(n)
        d = self.dev
        with torch.cuda.stream(self.s_up):
            if self._used[slot]:
                self.s_up.wait_event(self.e_run[slot])  # the slot's previous compute read d_in[slot]
            self.d_in[slot][:n].copy_((self.h_in[slot] if src is None else src)[:n], non_blocking=True)
            self.e_up[slot].record(self.s_up)
        with torch.cuda.stream(self.s_run):
            self.s_run.wait_event(self.e_up[slot])
            self.s_run.wait_event(self.e_down[slot])  # d_out[slot] was read by the slot's previous download
            check(d.lib.uwie_enhance_u8(d._ctx, _ptr(self.d_in[slot]), _ptr(self.d_out[slot]), None, n, self.H, self.W,
                                        ctypes.byref(self.p), _ptr(self.ws), self.ws.numel(),
                                        ctypes.c_void_p(self.s_run.cuda_stream)))
            self.e_run[slot].record(self.s_run)
        with torch.cuda.stream(self.s_down):
            self.s_down.wait_event(self.e_run[slot])
            self.h_out[slot][:n].copy_(self.d_out[slot][:n], non_blocking=True)
            self.e_down[slot].record(self.s_down)
        self._used[slot] = True
        self._pending.append((slot, n))

    def result(self):
        """Oldest submitted chunk, blocking until its download has finished."""
        slot, n = self._pending.pop(0)
        self.e_down[slot].synchronize()
        return self.h_out[slot][:n]

    # ------------------------------------------------------------------ iterator interface
    def run(self, chunks):
        i = 0
        for host in chunks:
            if len(self._pending) == self.depth - 1:
                yield self.result()
            t = torch.as_tensor(host)
            n = t.shape[0]
            if t.dtype != torch.uint8 or tuple(t.shape[1:]) != (self.H, self.W, 3) or n > self.chunk:
                raise ValueError(f"expected uint8 [<= {self.chunk}, {self.H}, {self.W}, 3], got {tuple(t.shape)} {t.dtype}")
            self.input_slot(i)[:n].copy_(t)
            self.submit_slot(i, n)
            i += 1
        while self._pending:
            yield self.result()
