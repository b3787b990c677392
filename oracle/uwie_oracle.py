"""CPU oracle for the enhancement hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of the reference's per-pixel pipeline, following the
reference's operation order and dtypes so that results are bit-identical to
what the reference computes on the same inputs:

* ``SixStrategyOracle``  follows ``six_stadigy.py:22-323`` (the hard-coded
  six-strategy surface plus cast detection/correction);
* ``DictStrategyOracle`` follows ``enhancement_strategies.py:13-508`` (the
  dict-parameterised ``apply_strategy`` surface);
* ``enhance_u8`` is the canonical ``enhance(u8) -> u8`` of SURVEY.md section 8
  (``six_stadigy.py:406,409,413,427,430``).

The seven OpenCV primitives come from ``oracle/cvref.c`` (own restatement,
see its header).  PINNING STATUS: every function that does not touch OpenCV
is pinned against the real reference imported with ``cv2`` stubbed
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``); the OpenCV-backed stages
are PARITY UNPINNED (no cv2 in this image, no golden outputs in the reference)
and are held only by hand-derived known-answer tests.  NumPy semantics are
those of NumPy 2.2.6 (float32 ``np.percentile`` arithmetic, NEP-50 promotion).

The callers either side of the path are restated at the end of the file:
``diff_enhance`` / ``extract_all_features`` (``vgg_16_UIE.py``; PINNED on outputs of the real module,
``oracle/gen_golden_vgg.py``) and ``quality_assessment`` (``quality_assessment.py``; PARITY UNPINNED, it runs on the
OpenCV restatements).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libcvref.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/cvref.c (gcc) if the shared object is missing."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _cv():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        vp, sz, i32, f64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
        lib.cvref_box_filter_f64.argtypes = [vp, vp, i32, i32, i32]
        lib.cvref_rgb2gray_u8.argtypes = [vp, vp, sz, i32]
        lib.cvref_rgb2lab_u8.argtypes = [vp, vp, sz]
        lib.cvref_lab2rgb_u8.argtypes = [vp, vp, sz]
        lib.cvref_clahe_u8.argtypes = [vp, vp, i32, i32, f64, i32, i32]
        lib.cvref_canny_u8.argtypes = [vp, vp, i32, i32, f64, f64]
        lib.cvref_equalize_hist_u8.argtypes = [vp, vp, sz]
        lib.cvref_lab_tables.argtypes = [vp] * 7
        for name in ("box_filter_f64", "rgb2gray_u8", "rgb2lab_u8", "lab2rgb_u8", "clahe_u8", "canny_u8",
                     "equalize_hist_u8", "lab_tables"):
            getattr(lib, "cvref_" + name).restype = None
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------
# OpenCV primitive wrappers (restated in cvref.c)
# --------------------------------------------------------------------------
GRAY_SHIFT_DEFAULT = 15  # OpenCV 4.x RGB2Gray<uchar>; 14 selects the older yuv_shift coefficients


def cv_box_filter_f64(plane, ksize):
    """cv2.boxFilter(plane, cv2.CV_64F, (ksize, ksize)) for a float64 HxW plane."""
    src = np.ascontiguousarray(plane, dtype=np.float64)
    dst = np.empty_like(src)
    _cv().cvref_box_filter_f64(_p(src), _p(dst), src.shape[0], src.shape[1], int(ksize))
    return dst


def cv_rgb2gray_u8(rgb, gray_shift=GRAY_SHIFT_DEFAULT):
    src = np.ascontiguousarray(rgb, dtype=np.uint8)
    dst = np.empty(src.shape[:2], np.uint8)
    _cv().cvref_rgb2gray_u8(_p(src), _p(dst), dst.size, int(gray_shift))
    return dst


def cv_rgb2lab_u8(rgb):
    src = np.ascontiguousarray(rgb, dtype=np.uint8)
    dst = np.empty_like(src)
    _cv().cvref_rgb2lab_u8(_p(src), _p(dst), src.size // 3)
    return dst


def cv_lab2rgb_u8(lab):
    src = np.ascontiguousarray(lab, dtype=np.uint8)
    dst = np.empty_like(src)
    _cv().cvref_lab2rgb_u8(_p(src), _p(dst), src.size // 3)
    return dst


def cv_clahe_u8(plane, clip_limit, tiles=(8, 8)):
    """cv2.createCLAHE(clipLimit, tileGridSize=tiles).apply(plane); tiles = (tilesX, tilesY)."""
    src = np.ascontiguousarray(plane, dtype=np.uint8)
    dst = np.empty_like(src)
    _cv().cvref_clahe_u8(_p(src), _p(dst), src.shape[0], src.shape[1], float(clip_limit), int(tiles[0]),
                         int(tiles[1]))
    return dst


def cv_canny_u8(plane, low, high):
    src = np.ascontiguousarray(plane, dtype=np.uint8)
    dst = np.empty_like(src)
    _cv().cvref_canny_u8(_p(src), _p(dst), src.shape[0], src.shape[1], float(low), float(high))
    return dst


def cv_rgb2hsv_u8(rgb):
    src = np.ascontiguousarray(rgb, dtype=np.uint8)
    dst = np.empty_like(src)
    _cv().cvref_rgb2hsv_u8(_p(src), _p(dst), src.size // 3)
    return dst


def cv_laplacian_f64(gray_f32):
    """cv2.Laplacian(gray, CV_64F) with the default ksize=1: kernel [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101,
    accumulated in float64 (filter2D).  PARITY UNPINNED."""
    g = np.pad(np.asarray(gray_f32, dtype=np.float64), 1, mode="reflect")
    return g[:-2, 1:-1] + g[2:, 1:-1] + g[1:-1, :-2] + g[1:-1, 2:] - 4.0 * g[1:-1, 1:-1]


def cv_equalize_hist_u8(plane):
    src = np.ascontiguousarray(plane, dtype=np.uint8)
    dst = np.empty_like(src)
    _cv().cvref_equalize_hist_u8(_p(src), _p(dst), src.size)
    return dst


def cv_lab_tables():
    """The integer LAB tables of cvref.c (used to cross-check the product's own tables)."""
    t = dict(gamma=np.zeros(256, np.uint16), invgamma=np.zeros(4096, np.uint16), cbrt=np.zeros(3072, np.uint16),
             ltoyf=np.zeros(512, np.int32), abtoxz=np.zeros(36864, np.int32), fwd=np.zeros(9, np.int32),
             inv=np.zeros(9, np.int32))
    _cv().cvref_lab_tables(_p(t["gamma"]), _p(t["invgamma"]), _p(t["cbrt"]), _p(t["ltoyf"]), _p(t["abtoxz"]),
                           _p(t["fwd"]), _p(t["inv"]))
    return t


# --------------------------------------------------------------------------
# u8 <-> float boundaries (six_stadigy.py:406,430; main.py:108,155)
# --------------------------------------------------------------------------
def normalise_u8(frame_u8):
    """six_stadigy.py:406 -- ``.astype(np.float32) / 255.0`` (true float32 division)."""
    return frame_u8.astype(np.float32) / 255.0


def quantise_u8(img):
    """six_stadigy.py:430 -- ``(img * 255).astype(np.uint8)`` (truncation toward zero)."""
    return (img * 255).astype(np.uint8)


# --------------------------------------------------------------------------
# cast detection / correction (six_stadigy.py:292-323)
# --------------------------------------------------------------------------
def classify_cast(img):
    """six_stadigy.py:292-302.  The per-channel mean is NumPy's float32 reduction over axes (0, 1)."""
    r, g, b = img.mean(axis=(0, 1))
    if g > r and g > b and (g - r) > 0.05:
        return "greenish"
    if b > r and b > g and (b - r) > 0.05:
        return "bluish"
    return "normal"


def correct_cast(img, kind):
    """six_stadigy.py:305-323 -- attenuate the dominant channel by 0.85, clip to [0, 1]."""
    channel = {"greenish": 1, "bluish": 2}.get(kind)
    if channel is None:
        return img  # same object, like the reference (six_stadigy.py:323)
    out = img.copy()
    out[:, :, channel] = out[:, :, channel] * 0.85
    return np.clip(out, 0, 1)


# --------------------------------------------------------------------------
# primitives shared by both surfaces
# --------------------------------------------------------------------------
def guided_filter(guide, src, ksize, eps):
    """six_stadigy.py:26-46 == enhancement_strategies.py:17-46 (``r`` is the box WIDTH)."""
    I = guide.astype(np.float64)
    p = src.astype(np.float64)
    m_I = cv_box_filter_f64(I, ksize)
    m_p = cv_box_filter_f64(p, ksize)
    m_Ip = cv_box_filter_f64(I * p, ksize)
    cov = m_Ip - m_I * m_p
    m_II = cv_box_filter_f64(I * I, ksize)
    var = m_II - m_I * m_I
    a = cov / (var + eps)
    b = m_p - a * m_I
    return cv_box_filter_f64(a, ksize) * I + cv_box_filter_f64(b, ksize)


def quality_score(block, gray_shift=GRAY_SHIFT_DEFAULT):
    """compute_Q: six_stadigy.py:116-157 == enhancement_strategies.py:147-188.

    Returns the float64 score and, for stage-level tests, its four terms.
    """
    rows, cols, _ = block.shape
    n = rows * cols
    ch = [block[:, :, c] for c in range(3)]
    tot = [np.sum(c) for c in ch]
    t_bright = (tot[0] + tot[1] + tot[2]) / (3 * n)
    t_colour = (tot[2] + tot[1] - 2 * tot[0]) / n
    var = [np.sum((c - np.mean(c)) ** 2) / n for c in ch]
    t_var = (var[0] + var[1] + var[2]) / 3
    gray = cv_rgb2gray_u8((block * 255).astype(np.uint8), gray_shift)
    edges = cv_canny_u8(gray, 50, 150)
    t_edge = np.sum(edges > 0) / n
    return t_bright + t_colour - t_var - t_edge, (t_bright, t_colour, t_var, t_edge)


def brightest_pixel(block):
    """get_brightest_pixel: six_stadigy.py:160-165 == enhancement_strategies.py:191-206."""
    s = np.sum(block, axis=2)
    iy, ix = np.unravel_index(np.argmax(s), s.shape)
    return block[iy, ix, :]


def atmospheric_light(img, min_size=1, gray_shift=GRAY_SHIFT_DEFAULT, trace=None):
    """Quadtree descent: six_stadigy.py:49-113 == enhancement_strategies.py:77-140.

    The reference's stack never holds more than one block, so this is a greedy
    walk: score the four quadrants, step into the first maximum, stop when a
    side is <= min_size, and return the brightest pixel of that one leaf.
    ``trace`` (a list) receives (y0, x0, rows, cols, scores) per level.
    """
    y0 = x0 = 0
    rows, cols = img.shape[:2]
    while not (rows <= min_size or cols <= min_size):
        mr, mc = rows // 2, cols // 2
        quads = [(y0, x0, mr, mc), (y0, x0 + mc, mr, cols - mc), (y0 + mr, x0, rows - mr, mc),
                 (y0 + mr, x0 + mc, rows - mr, cols - mc)]
        scores = [quality_score(img[qy:qy + qr, qx:qx + qc, :], gray_shift)[0] for qy, qx, qr, qc in quads]
        if trace is not None:
            trace.append((y0, x0, rows, cols, [float(s) for s in scores]))
        y0, x0, rows, cols = quads[int(np.argmax(scores))]
    return brightest_pixel(img[y0:y0 + rows, x0:x0 + cols, :])


# --------------------------------------------------------------------------
# six_stadigy.py surface
# --------------------------------------------------------------------------
class SixStrategyOracle:
    """Restatement of ``six_stadigy.EnhancementStrategies`` (six_stadigy.py:22-285)."""

    gray_shift = GRAY_SHIFT_DEFAULT

    # name, omega, ksize, eps, (L_low, L_high), clahe clip, wb percentile, gamma -- six_stadigy.py:230-285
    @classmethod
    def transmission_init(cls, img, A, omega):
        """six_stadigy.py:170-174 -- dark channel over the 3 channels only (1x1 patch), pre-filter clip."""
        dark = np.min(img / (A.reshape(1, 1, 3) + 1e-6), axis=2)
        return np.clip(1 - omega * dark, 0.1, 1.0)

    @classmethod
    def guide(cls, img):
        """six_stadigy.py:177 -- 8-bit gray of the truncated frame, as float64 in [0, 1]."""
        return cv_rgb2gray_u8((img * 255).astype(np.uint8), cls.gray_shift).astype(np.float64) / 255.0

    @classmethod
    def transmission(cls, img, A, omega, ksize, eps):
        """estimate_transmission: six_stadigy.py:168-180 (returns float64 HxW)."""
        t = guided_filter(cls.guide(img), cls.transmission_init(img, A, omega), ksize, eps)
        return np.clip(t, 0.1, 1.0)

    @staticmethod
    def restore(img, A, t):
        """restore_image: six_stadigy.py:183-188 (float64 arithmetic, float32 store)."""
        out = np.zeros_like(img)
        for c in range(3):
            out[:, :, c] = (img[:, :, c] - A[c]) / t + A[c]
        return np.clip(out, 0, 1)

    @staticmethod
    def stretch(img, lo_pct, hi_pct):
        """enhance_contrast: six_stadigy.py:191-199; white_balance (:211-219) is stretch(p, 100-p)."""
        out = np.zeros_like(img)
        for c in range(3):
            plane = img[:, :, c]
            lo = np.percentile(plane, lo_pct)
            hi = np.percentile(plane, hi_pct)
            out[:, :, c] = np.clip((plane - lo) / (hi - lo + 1e-6), 0, 1)
        return out

    @classmethod
    def white_balance(cls, img, pct=5):
        return cls.stretch(img, pct, 100 - pct)

    @staticmethod
    def clahe(img, clip=2.0):
        """apply_clahe: six_stadigy.py:202-208 (LAB lightness, 8x8 tiles, float32 result)."""
        lab = cv_rgb2lab_u8((img * 255).astype(np.uint8))
        lab[:, :, 0] = cv_clahe_u8(lab[:, :, 0], clip, (8, 8))
        return cv_lab2rgb_u8(lab).astype(np.float32) / 255.0

    @staticmethod
    def gamma(img, g=1.2):
        """gamma_correction: six_stadigy.py:222-224 -- x**g, no clip."""
        return np.power(img, g)

    @classmethod
    def dehaze(cls, img, omega, ksize, eps):
        A = atmospheric_light(img, 1, cls.gray_shift)
        return cls.restore(img, A, cls.transmission(img, A, omega, ksize, eps))

    @classmethod
    def strategy(cls, number, img):
        """strategy1..6: six_stadigy.py:230-285."""
        if number == 1:
            y = cls.stretch(cls.dehaze(img, 0.3, 20, 5e-1), 5, 98)
            return cls.gamma(cls.clahe(y, 3.0), 1.5)
        if number == 2:
            y = cls.stretch(cls.dehaze(img, 0.5, 15, 5e-1), 15, 95)
            return cls.clahe(y, 2.0)
        if number == 3:
            y = cls.stretch(cls.dehaze(img, 0.7, 10, 1e-1), 20, 85)
            return cls.white_balance(y, 2)
        if number == 4:
            y = cls.stretch(cls.clahe(img, 4.0), 10, 95)
            return cls.gamma(cls.white_balance(y, 3), 1.3)
        if number == 5:
            y = cls.stretch(cls.white_balance(img, 2), 15, 90)
            return cls.gamma(cls.clahe(y, 1.5), 1.2)
        if number == 6:
            return cls.gamma(cls.clahe(cls.stretch(img, 5, 98), 3.5), 1.4)
        raise ValueError(f"unknown strategy number {number}")


def enhance_u8(frame_u8, strategy=2, cast_correct=True):
    """Canonical ``enhance(u8 HxWx3) -> u8 HxWx3`` (six_stadigy.py:406-431)."""
    x = normalise_u8(frame_u8)
    if cast_correct:
        x = correct_cast(x, classify_cast(x))
    return quantise_u8(SixStrategyOracle.strategy(strategy, x))


# --------------------------------------------------------------------------
# enhancement_strategies.py surface
# --------------------------------------------------------------------------
class DictStrategyOracle:
    """Restatement of ``enhancement_strategies.EnhancementStrategies`` (:13-508)."""

    gray_shift = GRAY_SHIFT_DEFAULT

    @classmethod
    def atmosphere(cls, img, min_size=1):
        """enhancement_strategies.py:77-144 -- the leaf pixel tiled to HxWx3."""
        h, w = img.shape[:2]
        rgb = atmospheric_light(img, min_size, cls.gray_shift)
        return np.tile(rgb.reshape(1, 1, 3), (h, w, 1))

    @classmethod
    def transmission(cls, img, A, omega=0.95, r=15, eps=0.001):
        """enhancement_strategies.py:209-234 -- no clip before the filter, eps 1e-10 in the normalisation."""
        dark = np.min(img / (A + 1e-10), axis=2)
        t0 = 1 - omega * dark
        g = cv_rgb2gray_u8((img * 255).astype(np.uint8), cls.gray_shift).astype(np.float64) / 255.0
        return np.clip(guided_filter(g, t0, r, eps), 0.1, 1.0)

    @staticmethod
    def recover(img, t, A):
        """enhancement_strategies.py:237-249 -- result stays float64."""
        return np.clip((img - A) / np.expand_dims(t, axis=2) + A, 0, 1)

    @staticmethod
    def stretch(img, lo_pct=15, hi_pct=95):
        """color_enhancement: enhancement_strategies.py:252-273 (eps 1e-10, dtype of the input)."""
        out = np.zeros_like(img)
        for c in range(3):
            plane = img[:, :, c]
            lo = np.percentile(plane, lo_pct)
            hi = np.percentile(plane, hi_pct)
            out[:, :, c] = np.clip((plane - lo) / (hi - lo + 1e-10), 0, 1)
        return out

    @staticmethod
    def gamma(img, g=1.2):
        """enhancement_strategies.py:276-285 -- clip(x ** (1/g), 0, 1)."""
        return np.clip(np.power(img, 1.0 / g), 0, 1)

    @staticmethod
    def clahe(img, clip=2.0, tiles=(8, 8)):
        """enhancement_strategies.py:288-307 -- float64 result."""
        lab = cv_rgb2lab_u8((img * 255).astype(np.uint8))
        lab[:, :, 0] = cv_clahe_u8(lab[:, :, 0], clip, tiles)
        return cv_lab2rgb_u8(lab).astype(np.float64) / 255.0

    @staticmethod
    def hist_eq(img):
        """enhancement_strategies.py:331-345 -- per-channel equalizeHist, float64 result."""
        q = (img * 255).astype(np.uint8)
        out = np.zeros_like(q)
        for c in range(3):
            out[:, :, c] = cv_equalize_hist_u8(q[:, :, c])
        return out.astype(np.float64) / 255.0

    # per-strategy in-code defaults: enhancement_strategies.py:356-372,382-395,405-419,428-441,466-473
    _DEHAZE_DEFAULTS = {
        "strong_dehazing": (0.5, 15, 10, 95),
        "medium_dehazing": (0.6, 20, 15, 92),
        "light_enhancement": (0.4, 10, 15, 95),
    }

    @classmethod
    def _maybe_gamma(cls, img, params):
        if params.get("apply_gamma", False):
            return cls.gamma(img, params.get("gamma", 1.2))
        return img

    @classmethod
    def run(cls, img, name, params):
        """The body of one strategy (no error swallowing)."""
        if name in cls._DEHAZE_DEFAULTS:
            omega, r, lo, hi = cls._DEHAZE_DEFAULTS[name]
            A = cls.atmosphere(img, 1)
            t = cls.transmission(img, A, omega=params.get("omega", omega), r=params.get("guided_radius", r))
            y = cls.stretch(cls.recover(img, t, A), params.get("L_low", lo), params.get("L_high", hi))
            return cls._maybe_gamma(y, params)
        if name == "clahe_enhancement":
            y = cls.clahe(img, params.get("clip_limit", 2.0), params.get("tile_grid_size", (8, 8)))
            return cls._maybe_gamma(cls.stretch(y, params.get("L_low", 20), params.get("L_high", 85)), params)
        if name == "histogram_equalization":
            y = cls.stretch(cls.hist_eq(img), params.get("L_low", 10), params.get("L_high", 95))
            return cls._maybe_gamma(y, params)
        raise ValueError(f"unknown strategy: {name}")

    @classmethod
    def apply_strategy(cls, img, name, params):
        """enhancement_strategies.py:477-508 -- ValueError for unknown names, otherwise swallow-and-return-input."""
        if name not in cls._DEHAZE_DEFAULTS and name not in ("clahe_enhancement", "histogram_equalization"):
            raise ValueError(f"unknown strategy: {name}")
        try:
            return cls.run(img, name, params)
        except Exception as exc:  # noqa: BLE001 - mirrors the reference's blanket except (:506-508)
            print(f"strategy {name} failed: {exc}")
            return img


# Config.STRATEGIES values (config.py:28-75) -- the parameter sets the boundary must accept.
CONFIG_STRATEGIES = {
    "strong_dehazing": dict(omega=0.5, guided_radius=15, L_low=10, L_high=95, gamma=1.2, apply_gamma=True),
    "medium_dehazing": dict(omega=0.6, guided_radius=20, L_low=15, L_high=92, apply_gamma=True),
    "light_enhancement": dict(omega=0.4, guided_radius=10, L_low=15, L_high=95, apply_gamma=False),
    "clahe_enhancement": dict(clip_limit=2.0, tile_grid_size=(8, 8), apply_gamma=False),
    "histogram_equalization": dict(L_low=10, L_high=95),
}


# ---------------------------------------------------------------------------------------------------------------------
# vgg_16_UIE.DifferentiableEnhancement / extract_all_features.
# The reference evaluates these with PyTorch / NumPy on the CPU; the restatement uses the same libraries for the
# arithmetic (torch.sort, float32 tensor ops, torch.pow) so that it is bit-identical to the reference on the machine
# that runs both (tests/golden/vgg_stages.npz holds outputs of the real module, oracle/gen_golden_vgg.py).
def diff_enhance(img_bchw, params):
    """DifferentiableEnhancement.forward, vgg_16_UIE.py:32-55.  img: (B,3,H,W) float32; params: dict of (B,1) float32
    arrays with L_low, L_high and optionally omega, gamma.  Returns a float32 ndarray."""
    import torch

    img = torch.as_tensor(np.ascontiguousarray(img_bchw, dtype=np.float32))
    par = {k: torch.as_tensor(np.asarray(v, dtype=np.float32)).reshape(-1, 1) for k, v in params.items()}
    B, C, H, W = img.shape
    out = torch.zeros_like(img)
    for b in range(B):  # color_stretch_batch, vgg_16_UIE.py:57-93
        for c in range(C):
            ch = img[b, c]
            srt, _ = torch.sort(ch.flatten())
            n = len(srt)
            lo_i = max(0, min(int((par["L_low"][b].item() / 100.0) * n), n - 1))
            hi_i = max(0, min(int((par["L_high"][b].item() / 100.0) * n), n - 1))
            rng = srt[hi_i] - srt[lo_i] + 1e-8
            out[b, c] = torch.clamp((ch - srt[lo_i]) / rng, 0, 1)
    if "omega" in par:  # dehaze_batch, vgg_16_UIE.py:95-117
        omega = par["omega"].view(-1, 1, 1, 1)
        dark = torch.min(out, dim=1, keepdim=True)[0]
        t = torch.clamp(1 - omega * dark, 0.1, 1.0)
        out = torch.clamp((out - 0.6) / t + 0.6, 0, 1)
    if "gamma" in par:  # gamma_correction, vgg_16_UIE.py:119-128
        out = torch.pow(out + 1e-8, par["gamma"].view(-1, 1, 1, 1))
    return torch.clamp(out, 0, 1).numpy()


def diff_enhance_image(img_hwc, params):
    """EnhancementPredictor.enhance_image with explicit parameters, use_trained_model.py:83-111."""
    x = np.asarray(img_hwc, dtype=np.float32)
    par = {k: np.array([[float(params[k])]], np.float32) for k in ("omega", "gamma", "L_low", "L_high")}
    out = diff_enhance(np.ascontiguousarray(x.transpose(2, 0, 1))[None], par)[0].transpose(1, 2, 0)
    return np.clip(out, 0.0, 1.0)


def extract_all_features(img):
    """vgg_16_UIE.extract_all_features, vgg_16_UIE.py:435-466 (with _ensure_float01, :417-427)."""
    img = np.asarray(img)
    if img.dtype == np.uint8:
        img = img.astype(np.float32) / 255.0
    else:
        img = img.astype(np.float32)
        if img.max() > 1.0:
            img = img / 255.0
    feats = []
    for c in range(3):
        ch = img[:, :, c]
        feats.extend([float(np.mean(ch)), float(np.std(ch)), float(np.min(ch)), float(np.max(ch)), float(np.median(ch))])
    feats.extend([float(np.mean(img)), float(np.std(img)), float(np.mean(img ** 2))])
    while len(feats) < 79:
        feats.append(0.0)
    return np.array(feats[:79], dtype=np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# quality_assessment.QualityAssessment: the eight no-reference scores and their weighted
# sum, restated line by line on the oracle's OpenCV restatements.  skimage.measure.shannon_entropy is restated from
# its published definition: -sum(p * log2(p)) over the relative frequencies of the distinct values (scipy.stats.entropy).
QUALITY_KEYS = ("contrast", "sharpness", "entropy", "saturation", "brightness", "edge_density", "colorfulness", "naturalness")
QUALITY_WEIGHTS = {"contrast": 0.20, "sharpness": 0.20, "entropy": 0.15, "saturation": 0.15, "brightness": 0.10,
                   "edge_density": 0.10, "colorfulness": 0.05, "naturalness": 0.05}  # quality_assessment.py:229-238


def quality_scores(img, gray_shift=GRAY_SHIFT_DEFAULT):
    """QualityAssessment.assess_* (quality_assessment.py:15-212) of a float RGB image in [0, 1]; returns the dict."""
    img = np.asarray(img)
    u8 = (img * 255).astype(np.uint8)
    gray_u8 = cv_rgb2gray_u8(u8, gray_shift)
    gray = gray_u8.astype(np.float32) / 255.0
    sc = {}
    sc["contrast"] = np.clip(np.std(gray) / 0.5 * 100, 0, 100)                                     # :25-31
    sc["sharpness"] = np.clip(np.var(cv_laplacian_f64(gray)) / 0.5 * 100, 0, 100)                  # :45-52
    _, counts = np.unique(gray, return_counts=True)
    pk = counts / counts.sum()
    entropy = -np.sum(pk * np.log(pk)) / np.log(2)
    sc["entropy"] = np.clip((entropy - 4) / 4 * 100, 0, 100)                                       # :66-73
    hsv = cv_rgb2hsv_u8(u8).astype(np.float32) / 255.0
    sc["saturation"] = np.clip(np.mean(hsv[:, :, 1]) * 100, 0, 100)                                # :87-94
    L = cv_rgb2lab_u8(u8).astype(np.float32)[:, :, 0]
    sc["brightness"] = 100 - np.clip(abs(np.mean(L) - 128) / 128 * 100, 0, 100)                    # :108-117
    edges = cv_canny_u8(gray_u8, 50, 150)
    sc["edge_density"] = np.clip(np.sum(edges > 0) / edges.size / 0.2 * 100, 0, 100)               # :131-140
    R, G, B = img[:, :, 0], img[:, :, 1], img[:, :, 2]
    rg, yb = R - G, 0.5 * (R + G) - B
    colorfulness = np.sqrt(np.std(rg) ** 2 + np.std(yb) ** 2) + 0.3 * np.sqrt(np.mean(rg) ** 2 + np.mean(yb) ** 2)
    sc["colorfulness"] = np.clip(colorfulness / 0.5 * 100, 0, 100)                                 # :154-175
    unnatural = (np.sum(hsv[:, :, 1] > 0.9) / hsv[:, :, 1].size + np.sum(gray < 0.1) / gray.size
                 + np.sum(gray > 0.9) / gray.size)
    sc["naturalness"] = 100 - np.clip(unnatural * 200, 0, 100)                                     # :189-205
    return sc


def quality_assessment(img, weights=None, gray_shift=GRAY_SHIFT_DEFAULT):
    """comprehensive_assessment (quality_assessment.py:215-286): (total, scores)."""
    weights = QUALITY_WEIGHTS if weights is None else weights
    scores = quality_scores(img, gray_shift)
    return sum(scores[k] * weights.get(k, 0) for k in scores), scores
