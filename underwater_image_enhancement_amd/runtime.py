"""Device runtime: one libuwie context per GPU, torch tensors as the HBM containers.

PyTorch is plumbing here (device memory, streams); every enhancement operation is a
hand-written HIP kernel reached through the C ABI in include/uwie.h.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import UwieParams, check

_TRACE_DTYPE = np.dtype([("y0", "<i4"), ("x0", "<i4"), ("rows", "<i4"), ("cols", "<i4"), ("score", "<f8", (4,))])


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Device:
    """A libuwie context bound to one MI355X.  Not thread-safe; use one per host thread / stream."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.UwieError("no ROCm device visible: the enhancement path has no CPU fallback")
        self.lib = _lib.load()
        self.index = int(device)
        self.torch_device = torch.device("cuda", self.index)
        torch.cuda.set_device(self.index)
        handle = ctypes.c_void_p()
        check(self.lib.uwie_create(self.index, ctypes.byref(handle)))
        self._ctx = handle
        self._workspace = None

    def close(self):
        if getattr(self, "_ctx", None):
            torch.cuda.synchronize(self.index)
            self.lib.uwie_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.index).cuda_stream)

    def workspace(self, nbytes: int):
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = None
            self._workspace = torch.empty(int(nbytes), dtype=torch.uint8, device=self.torch_device)
        return self._workspace

    def params(self, surface: int, strategy: int, **overrides) -> UwieParams:
        p = UwieParams()
        check(self.lib.uwie_params_init(ctypes.byref(p), surface, strategy))
        for key, value in overrides.items():
            if not hasattr(p, key):
                raise KeyError(key)
            setattr(p, key, value)
        return p

    def workspace_for(self, B, H, W, p: UwieParams | None = None):
        n = self.lib.uwie_workspace_bytes_ctx(self._ctx, B, H, W, ctypes.byref(p) if p is not None else None)
        if n == 0:
            raise _lib.UwieError("batch/H/W out of range")
        return self.workspace(n)

    def tensor(self, array, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(array))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.torch_device)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.torch_device)

    @staticmethod
    def _bhw(img):
        assert img.dim() == 4 and img.shape[-1] == 3 and img.is_contiguous(), "expected contiguous [B,H,W,3]"
        return int(img.shape[0]), int(img.shape[1]), int(img.shape[2])

    # ------------------------------------------------------------------ route selectors (uwie_set_tuning)
    def tune(self, **selectors):
        """Set route selectors of this context (include/uwie.h: gf_pipe, gf_split, gf_bands, select_generic, restore_store,
        lin_predict3, lin_cap, lin_no_predict, lin_predict_shift, q_hist, streams, canny_prepass, entry_fuse ...).  The selection / storage / quadtree
        selectors give the same bytes on every route, the gf_* ones the same transmission to 1e-11 (uwie.h); tests force the
        fallback routes with it."""
        for name, value in selectors.items():
            check(self.lib.uwie_set_tuning(self._ctx, name.encode(), int(value)))

    def tuning(self, **selectors):
        """Context manager: ``with dev.tuning(lin_cap=16): ...`` sets the selectors and puts the old values back."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {}
            for name in selectors:
                v = ctypes.c_int()
                check(self.lib.uwie_get_tuning(self._ctx, name.encode(), ctypes.byref(v)))
                old[name] = v.value
            self.tune(**selectors)
            try:
                yield self
            finally:
                self.tune(**old)

        return scope()

    # ------------------------------------------------------------------ device-side self checks (uwie_device_status)
    def check_status(self):
        """Wait for the current stream and raise ``UwieError`` if a kernel found one of its invariants violated since the last
        check (include/uwie.h: UWIE_E_DEVICE) -- the results of those calls are not valid.  ``enhance`` & co. call this when
        they copy results back to the host (they synchronise there anyway); callers that keep tensors on the device call it
        when they synchronise."""
        check(self.lib.uwie_device_status(self._ctx, self.stream(), None))

    # ------------------------------------------------------------------ per-kernel timing (HIP events on the launch stream)
    def profile(self, on: bool, only: str | None = None):
        """Per-kernel HIP-event timing on/off; ``only`` restricts it to one kernel name (an event pair per launch costs
        stream time, so a whole-job measurement records the one kernel it reports)."""
        check(self.lib.uwie_profile_filter(self._ctx, only.encode() if only else None))
        check(self.lib.uwie_profile_enable(self._ctx, int(bool(on))))

    def profile_rows(self):
        """Synchronise and return {kernel name: (total ms, launches)} recorded since the last call."""
        n = self.lib.uwie_profile_collect(self._ctx)
        if n < 0:
            check(n)
        rows = {}
        name, ms, calls = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
        for i in range(n):
            check(self.lib.uwie_profile_row(self._ctx, i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(calls)))
            rows[name.value.decode()] = (ms.value, calls.value)
        return rows

    # ------------------------------------------------------------------ whole pipeline
    def enhance_u8(self, frames, p: UwieParams, want_float: bool = False):
        """frames: uint8 cuda tensor [B,H,W,3] -> (uint8 [B,H,W,3], float32 [B,H,W,3] or None)."""
        B, H, W = self._bhw(frames)
        assert frames.dtype == torch.uint8
        ws = self.workspace_for(B, H, W, p)
        out = self.empty((B, H, W, 3), torch.uint8)
        outf = self.empty((B, H, W, 3), torch.float32) if want_float else None
        check(self.lib.uwie_enhance_u8(self._ctx, _ptr(frames), _ptr(out), _ptr(outf), B, H, W, ctypes.byref(p),
                                       _ptr(ws), ws.numel(), self.stream()))
        return out, outf

    def enhance_u8_f64(self, frames, p: UwieParams):
        """Dict surface: uint8 cuda tensor [B,H,W,3] -> (uint8 [B,H,W,3], float64 [B,H,W,3]): the reference's float64 image."""
        B, H, W = self._bhw(frames)
        assert frames.dtype == torch.uint8
        ws = self.workspace_for(B, H, W, p)
        out = self.empty((B, H, W, 3), torch.uint8)
        outf = self.empty((B, H, W, 3), torch.float64)
        check(self.lib.uwie_enhance_u8_f64(self._ctx, _ptr(frames), _ptr(out), _ptr(outf), B, H, W, ctypes.byref(p),
                                           _ptr(ws), ws.numel(), self.stream()))
        return out, outf

    def enhance_float(self, img, p: UwieParams, want: str = "native"):
        """General (not u8-derived) float images [B,H,W,3], float32 (either surface) or float64 (dict surface): returns
        (uint8 [B,H,W,3], float image): float32 for the six_stadigy surface, float64 for the dict surface."""
        assert img.dim() == 4 and img.shape[-1] == 3 and img.is_contiguous(), "expected contiguous [B,H,W,3]"
        B, H, W = (int(v) for v in img.shape[:3])
        six = p.surface == _lib.SURFACE_SIX
        eb = 4 if img.dtype == torch.float32 else 8
        ws = self.workspace(self.lib.uwie_workspace_bytes_float(B, H, W, ctypes.byref(p), eb))
        out = self.empty((B, H, W, 3), torch.uint8)
        if img.dtype == torch.float32:
            of32 = self.empty((B, H, W, 3), torch.float32) if six else None
            of64 = None if six else self.empty((B, H, W, 3), torch.float64)
            check(self.lib.uwie_enhance_f32(self._ctx, _ptr(img), _ptr(out), _ptr(of32), _ptr(of64), B, H, W, ctypes.byref(p),
                                            _ptr(ws), ws.numel(), self.stream()))
            return out, (of32 if six else of64)
        assert img.dtype == torch.float64
        of64 = self.empty((B, H, W, 3), torch.float64)
        check(self.lib.uwie_enhance_f64(self._ctx, _ptr(img), _ptr(out), _ptr(of64), B, H, W, ctypes.byref(p), _ptr(ws),
                                        ws.numel(), self.stream()))
        return out, of64

    def cast_classify_f32(self, img):
        B, H, W = (int(v) for v in img.shape[:3])
        kind = self.empty((B,), torch.int32)
        mean = self.empty((B, 3), torch.float32)
        check(self.lib.uwie_cast_classify_f32(self._ctx, _ptr(img), B, H, W, _ptr(kind), _ptr(mean), self.stream()))
        return kind, mean

    def color_correct_f32(self, img, kind):
        B, H, W = (int(v) for v in img.shape[:3])
        out = self.empty((B, H, W, 3), torch.float32)
        check(self.lib.uwie_color_correct_f32(self._ctx, _ptr(img), _ptr(kind), _ptr(out), B, H, W, self.stream()))
        return out

    def enhance_all_u8(self, frames, cast_correct: bool = True):
        """frames: uint8 cuda tensor [B,H,W,3] -> (uint8 [6,B,H,W,3] = strategies 1..6, int32 [B] cast kinds)."""
        B, H, W = self._bhw(frames)
        assert frames.dtype == torch.uint8
        p6 = (UwieParams * 6)()
        for k in range(6):
            p6[k] = self.params(_lib.SURFACE_SIX, k + 1, cast_correct=int(bool(cast_correct)))
        ws = self.workspace(self.lib.uwie_workspace_bytes_all(B, H, W, ctypes.cast(p6, ctypes.c_void_p)))
        out = self.empty((6, B, H, W, 3), torch.uint8)
        kind = self.empty((B,), torch.int32)
        check(self.lib.uwie_enhance_all_u8(self._ctx, _ptr(frames), _ptr(out), _ptr(kind), B, H, W,
                                           ctypes.cast(p6, ctypes.c_void_p), _ptr(ws), ws.numel(), self.stream()))
        return out, kind

    def diff_enhance_f32(self, img, params, planar: bool, has_omega: bool = True, has_gamma: bool = True):
        """img: float32 cuda tensor [B,3,H,W] (planar) or [B,H,W,3]; params: float32 [B,4] = L_low, L_high, omega, gamma."""
        assert img.dtype == torch.float32 and params.dtype == torch.float32 and img.dim() == 4
        B = img.shape[0]
        H, W = (img.shape[2], img.shape[3]) if planar else (img.shape[1], img.shape[2])
        assert img.shape[1 if planar else 3] == 3 and tuple(params.shape) == (B, 4)
        img, params = img.contiguous(), params.contiguous()
        ws = self.workspace_for(B, H, W)
        out = self.empty(tuple(img.shape), torch.float32)
        check(self.lib.uwie_diff_enhance_f32(self._ctx, _ptr(img), _ptr(out), B, H, W, int(planar), _ptr(params),
                                             (1 if has_omega else 0) | (2 if has_gamma else 0), _ptr(ws), ws.numel(),
                                             self.stream()))
        return out

    def extract_features_u8(self, frames):
        """frames: uint8 cuda tensor [B,H,W,3] -> float32 [B,79] (vgg_16_UIE.extract_all_features per frame)."""
        B, H, W = self._bhw(frames)
        assert frames.dtype == torch.uint8
        ws = self.workspace_for(B, H, W)
        out = self.empty((B, 79), torch.float32)
        check(self.lib.uwie_extract_features_u8(self._ctx, _ptr(frames), _ptr(out), B, H, W, _ptr(ws), ws.numel(),
                                                self.stream()))
        return out

    def quality_scores(self, frames_u8, frames_f32=None, weights=None, gray_shift: int = 15):
        """frames_u8: uint8 cuda [B,H,W,3] (the quantised image); frames_f32: optional float32 cuda [B,H,W,3];
        weights: optional 8 floats.  Returns float64 [B,9]: the eight scores (QUALITY_KEYS order) and the total."""
        B, H, W = self._bhw(frames_u8)
        assert frames_u8.dtype == torch.uint8
        if frames_f32 is not None:
            assert frames_f32.dtype == torch.float32 and tuple(frames_f32.shape) == (B, H, W, 3)
            frames_f32 = frames_f32.contiguous()
        ws = self.workspace_for(B, H, W)
        out = self.empty((B, 9), torch.float64)
        w = (ctypes.c_double * 8)(*[float(x) for x in weights]) if weights is not None else None
        check(self.lib.uwie_quality_scores(self._ctx, _ptr(frames_u8), _ptr(frames_f32), B, H, W, int(gray_shift), w,
                                           _ptr(out), _ptr(ws), ws.numel(), self.stream()))
        return out

    def select_best_u8(self, frames, plist, weights=None, want_all: bool = False):
        """main.py:118-146 on the device (uwie_select_best_u8): frames uint8 cuda [B,H,W,3], plist a list of UwieParams.
        Returns (best index int32 [B], best image uint8 [B,H,W,3], scores float64 [n,B,9], all outputs [n,B,H,W,3] or None)."""
        B, H, W = self._bhw(frames)
        assert frames.dtype == torch.uint8
        n = len(plist)
        arr = (UwieParams * n)(*plist)
        nbytes = self.lib.uwie_workspace_bytes_select(B, H, W, arr, n, int(want_all))
        if nbytes == 0:
            raise _lib.UwieError("select_best: batch/H/W or strategy count out of range")
        ws = self.workspace(nbytes)
        best = self.empty((B,), torch.int32)
        img = self.empty((B, H, W, 3), torch.uint8)
        scores = self.empty((n, B, 9), torch.float64)
        every = self.empty((n, B, H, W, 3), torch.uint8) if want_all else None
        w = (ctypes.c_double * 8)(*[float(x) for x in weights]) if weights is not None else None
        check(self.lib.uwie_select_best_u8(self._ctx, _ptr(frames), B, H, W, arr, n, w, _ptr(img), _ptr(best), _ptr(scores),
                                           _ptr(every), _ptr(ws), ws.numel(), self.stream()))
        return best, img, scores, every

    # ------------------------------------------------------------------ stages
    def cast_classify(self, frames):
        B, H, W = self._bhw(frames)
        ws = self.workspace_for(B, H, W)
        kind = self.empty((B,), torch.int32)
        mean = self.empty((B, 3), torch.float32)
        check(self.lib.uwie_cast_classify(self._ctx, _ptr(frames), B, H, W, _ptr(kind), _ptr(mean), _ptr(ws),
                                          ws.numel(), self.stream()))
        return kind, mean

    def normalise_correct(self, frames, kind=None):
        B, H, W = self._bhw(frames)
        out = self.empty((B, H, W, 3), torch.float32)
        check(self.lib.uwie_normalise_correct(self._ctx, _ptr(frames), _ptr(kind), _ptr(out), B, H, W, self.stream()))
        return out

    def atmospheric_light(self, frames, kind=None, p: UwieParams | None = None, trace: bool = False, want_gray: bool = False):
        """``want_gray``: also return the gray plane the call wrote on its way (uwie_atmospheric_light keeps it in the first
        B*H*W bytes of its workspace) -- tests compare it with ``transmission_init``'s."""
        B, H, W = self._bhw(frames)
        p = p or self.params(_lib.SURFACE_SIX, 2)
        ws = self.workspace_for(B, H, W, p)
        A = self.empty((B, 3), torch.float32)
        tr = torch.zeros((B, 32, _TRACE_DTYPE.itemsize), dtype=torch.uint8, device=self.torch_device) if trace else None
        check(self.lib.uwie_atmospheric_light(self._ctx, _ptr(frames), _ptr(kind), B, H, W, ctypes.byref(p), _ptr(A),
                                              _ptr(tr), _ptr(ws), ws.numel(), self.stream()))
        out = (A, tr.cpu().numpy().view(_TRACE_DTYPE).reshape(B, 32)) if trace else A
        if want_gray:
            gray = ws[: B * H * W].view(B, H, W).clone()
            return (*out, gray) if trace else (out, gray)
        return out

    def transmission_init(self, frames, A, kind=None, p: UwieParams | None = None):
        B, H, W = self._bhw(frames)
        p = p or self.params(_lib.SURFACE_SIX, 2)
        t0 = self.empty((B, H, W), torch.float32)
        gray = self.empty((B, H, W), torch.uint8)
        check(self.lib.uwie_transmission_init(self._ctx, _ptr(frames), _ptr(kind), _ptr(A), B, H, W, ctypes.byref(p),
                                              _ptr(t0), _ptr(gray), self.stream()))
        return t0, gray

    def box_filter_f64(self, planes, ksize):
        B, H, W = (int(v) for v in planes.shape)
        assert planes.dtype == torch.float64 and planes.is_contiguous()
        ws = self.workspace_for(B, H, W)
        out = torch.empty_like(planes)
        check(self.lib.uwie_box_filter_f64(self._ctx, _ptr(planes), _ptr(out), B, H, W, int(ksize), _ptr(ws), ws.numel(),
                                           self.stream()))
        return out

    def guided_filter(self, gray, t0, ksize, eps, exact=True):
        B, H, W = (int(v) for v in gray.shape)
        ws = self.workspace_for(B, H, W)
        t = self.empty((B, H, W), torch.float64)
        check(self.lib.uwie_guided_filter(self._ctx, _ptr(gray), _ptr(t0), B, H, W, int(ksize), float(eps), int(exact), _ptr(t),
                                          _ptr(ws), ws.numel(), self.stream()))
        return t

    def restore(self, frames, A, t, kind=None):
        B, H, W = self._bhw(frames)
        out = self.empty((B, H, W, 3), torch.float32)
        check(self.lib.uwie_restore(self._ctx, _ptr(frames), _ptr(kind), _ptr(A), _ptr(t), B, H, W, _ptr(out),
                                    self.stream()))
        return out

    def percentiles_f32(self, img, q_percent):
        B, H, W = self._bhw(img)
        assert img.dtype == torch.float32
        q = (ctypes.c_double * len(q_percent))(*[float(v) for v in q_percent])
        ws = self.workspace_for(B, H, W)
        out = self.empty((B, 3, len(q_percent)), torch.float32)
        check(self.lib.uwie_percentiles_f32(self._ctx, _ptr(img), B, H, W, q, len(q_percent), _ptr(out), _ptr(ws),
                                            ws.numel(), self.stream()))
        return out

    def stretch_f32(self, img, lo, hi):
        B, H, W = self._bhw(img)
        ws = self.workspace_for(B, H, W)
        out = torch.empty_like(img)
        check(self.lib.uwie_stretch_f32(self._ctx, _ptr(img), _ptr(out), B, H, W, float(lo), float(hi), _ptr(ws),
                                        ws.numel(), self.stream()))
        return out

    def gamma_f32(self, img, g, mode=1):
        out = torch.empty_like(img)
        check(self.lib.uwie_gamma_f32(self._ctx, _ptr(img), _ptr(out), img.numel(), float(g), int(mode), self.stream()))
        return out

    def clahe_f32(self, img, clip, tiles=(8, 8)):
        B, H, W = self._bhw(img)
        ws = self.workspace_for(B, H, W)
        out = torch.empty_like(img)
        check(self.lib.uwie_clahe_f32(self._ctx, _ptr(img), _ptr(out), B, H, W, float(clip), int(tiles[0]),
                                      int(tiles[1]), _ptr(ws), ws.numel(), self.stream()))
        return out

    def rgb2gray_u8(self, rgb, gray_shift=15):
        out = self.empty(rgb.shape[:-1], torch.uint8)
        check(self.lib.uwie_rgb2gray_u8(self._ctx, _ptr(rgb), _ptr(out), out.numel(), int(gray_shift), self.stream()))
        return out

    def rgb2lab_u8(self, rgb):
        out = torch.empty_like(rgb)
        check(self.lib.uwie_rgb2lab_u8(self._ctx, _ptr(rgb), _ptr(out), rgb.numel() // 3, self.stream()))
        return out

    def lab2rgb_u8(self, lab):
        out = torch.empty_like(lab)
        check(self.lib.uwie_lab2rgb_u8(self._ctx, _ptr(lab), _ptr(out), lab.numel() // 3, self.stream()))
        return out

    def clahe_u8(self, planes, clip, tiles=(8, 8)):
        B, H, W = (int(v) for v in planes.shape)
        ws = self.workspace_for(B, H, W)
        out = torch.empty_like(planes)
        check(self.lib.uwie_clahe_u8(self._ctx, _ptr(planes), _ptr(out), B, H, W, float(clip), int(tiles[0]),
                                     int(tiles[1]), _ptr(ws), ws.numel(), self.stream()))
        return out

    def canny_u8(self, gray, low=50, high=150):
        B, H, W = (int(v) for v in gray.shape)
        ws = self.workspace_for(B, H, W)
        out = torch.empty_like(gray)
        check(self.lib.uwie_canny_u8(self._ctx, _ptr(gray), _ptr(out), B, H, W, int(low), int(high), _ptr(ws),
                                     ws.numel(), self.stream()))
        return out

    def equalize_hist_u8(self, planes):
        B, H, W = (int(v) for v in planes.shape)
        ws = self.workspace_for(B, H, W)
        out = torch.empty_like(planes)
        check(self.lib.uwie_equalize_hist_u8(self._ctx, _ptr(planes), _ptr(out), B, H, W, _ptr(ws), ws.numel(),
                                             self.stream()))
        return out


_devices: dict[int, Device] = {}


def get_device(index: int | None = None) -> Device:
    if index is None:
        index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if index not in _devices:
        _devices[index] = Device(index)
    return _devices[index]
