"""Host-side mirror of the reference's enhancement API, driving the HIP kernels.

Reference surfaces mirrored here (same names, argument meaning and error behaviour):

* ``enhance(frame_u8)``                         -- the canonical ``enhance(img) -> img`` of SURVEY.md section 8:
  ``six_stadigy.py:406`` (u8 -> float32/255), ``:409-413`` (cast detection + correction),
  ``:427`` (``strategyN``), ``:430`` (``(y*255).astype(uint8)``);
* ``SixStrategies.strategy1_strong_dehazing`` ... ``strategy6_histogram_eq``, ``detect_image_type``,
  ``color_correction``                           -- ``six_stadigy.py:230-285,292-323``;
* ``EnhancementStrategies.apply_strategy(img, name, params)`` -- ``enhancement_strategies.py:477-508``.

Float images: a u8-derived image (``u8.astype(float32)/255`` possibly followed by ``color_correction``), which is what the
reference's pipelines pass (``six_stadigy.py:406``, ``main.py:108``), is recognised and takes the fast path that starts
from the u8 frame.  Any other float image in ``[0, 1]`` -- the reference's own harness inputs are ``np.random.rand``
(``enhancement_strategies.py:516``, ``example_usage.py:27,44,112``) -- takes the general float path
(``uwie_enhance_f32`` / ``uwie_enhance_f64``: same arithmetic, pixel values read from the float image, unfused):
float32 on both surfaces, float64 on the dict surface.  What is left raises ``UnsupportedInputError`` (float64 images on
the six_stadigy surface, whose functions are written for float32 frames; other dtypes) -- also from ``apply_strategy``,
whose swallow-and-return-the-input convention (ES:503-508) is for failures INSIDE a strategy: an input this build cannot
process is never reported as a successful pass-through.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .runtime import Device, get_device

class UnsupportedInputError(ValueError):
    """The float image is not u8-derived: this build has no device path for it (see the module docstring)."""


_F255 = np.float32(255.0)
_ATTEN = np.float32(0.85)


def _as_batch_u8(frames, dev: Device):
    """numpy/torch, [H,W,3] or [B,H,W,3] uint8 -> (cuda uint8 [B,H,W,3], was_numpy, was_single)."""
    was_numpy = isinstance(frames, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(frames)) if was_numpy else frames
    if t.dtype != torch.uint8:
        raise TypeError(f"expected uint8 frames, got {t.dtype}")
    single = t.dim() == 3
    if single:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[-1] != 3:
        raise ValueError(f"expected [H,W,3] or [B,H,W,3], got {tuple(t.shape)}")
    if t.numel() == 0:
        raise ValueError("empty image")
    return t.to(dev.torch_device).contiguous(), was_numpy, single


def _finish(t, was_numpy, single, dev: Device | None = None):
    if single:
        t = t[0]
    if not was_numpy:
        return t
    out = t.cpu().numpy()
    if dev is not None:
        dev.check_status()  # (the copy above synchronised already: this only reads the device's status word)
    return out


def enhance(frames, strategy: int = 2, cast_correct: bool = True, device: int | None = None, return_float: bool = False,
            **overrides):
    """Canonical ``enhance(u8 RGB) -> u8 RGB`` (six_stadigy.py:406-431) for one frame or a batch.

    ``frames``: uint8 ``[H,W,3]`` / ``[B,H,W,3]``, NumPy (copied to HBM and back) or a torch ROCm tensor (stays on
    the device).  ``strategy`` selects ``strategy1..6`` (default 2, medium dehazing).  ``overrides`` set
    ``uwie_params`` fields (e.g. ``omega=0.6, gf_ksize=20``).
    """
    dev = get_device(device)
    batch, was_numpy, single = _as_batch_u8(frames, dev)
    p = dev.params(_lib.SURFACE_SIX, int(strategy), cast_correct=int(bool(cast_correct)), **overrides)
    out, outf = dev.enhance_u8(batch, p, want_float=return_float)
    return _finish(outf if return_float else out, was_numpy, single, dev)


# ------------------------------------------------------------------ the batch driver's fan-out (six_stadigy.py:330-520)
# (name, description) of the driver's strategy table, six_stadigy.py:344-351
DRIVER_STRATEGIES = (
    ("strong_dehazing", "強力去霧"), ("medium_dehazing", "中度去霧"), ("light_dehazing", "輕度去霧"),
    ("clahe_enhancement", "CLAHE增強"), ("white_balance", "白平衡主導"), ("histogram_eq", "直方圖均衡"),
)
CAST_NAMES = ("normal", "greenish", "bluish")


def enhance_all(frames, cast_correct: bool = True, device: int | None = None):
    """All six strategies of every frame in one call: one cast detection and one atmospheric-light quadtree per frame
    (strategies 1-3 share it), like the inner loop of ``process_all_images_all_strategies`` (six_stadigy.py:398-431).

    Returns ``(outputs, image_types)``: ``outputs[name]`` is the uint8 batch of that strategy (keys in the driver's
    order, see ``DRIVER_STRATEGIES``), ``image_types`` the list of ``"normal" / "greenish" / "bluish"`` per frame.
    """
    dev = get_device(device)
    batch, was_numpy, single = _as_batch_u8(frames, dev)
    out, kind = dev.enhance_all_u8(batch, cast_correct)
    types = [CAST_NAMES[int(k)] for k in kind.cpu().tolist()]
    outs = {name: _finish(out[i], was_numpy, single, dev) for i, (name, _) in enumerate(DRIVER_STRATEGIES)}
    return outs, types


def process_batch(frames, filenames=None, device: int | None = None, compute=None, compute_one=None):
    """Host mirror of the driver's bookkeeping (six_stadigy.py:369-520) for frames that are already decoded: returns
    ``(outputs, log_rows, stats)`` where ``log_rows`` has one dict per (image, strategy) with the driver's columns
    (``filename, image_type, strategy, strategy_desc, status, output_path, processing_time``) and ``stats`` its counters.
    File I/O (glob / imread / imwrite / CSV) stays with the caller: ``output_path`` is empty for a success (the caller
    fills it in when it writes the file) and ``"Error: <first 50 characters>"`` for a failure, as in S6:464-478.

    The whole batch goes through ONE fused call (``enhance_all``: one cast detection and one quadtree per frame).  Failure
    handling follows the reference's two ``try`` levels (S6:395,424): if the fused call raises, every image is retried on
    its own; if an image's fused call raises, its six strategies run one by one, and a strategy that raises becomes a row
    with ``status='failed'``, ``processing_time='N/A'`` and counts in ``failed_outputs`` while the other five still succeed
    (S6:464-480); an image none of whose strategies succeeds counts in ``failed_images`` (S6:484-488), and one whose cast
    detection itself fails gets no rows at all (S6:508-510).  ``processing_time`` of a success is the time since the image
    started (S6:457): the image's share of the fused call, or the running time of the one-by-one fallback.
    ``outputs[name]`` is a list with one uint8 frame (or ``None`` for a failed output) per image.
    ``frames``: ``[B,H,W,3]`` uint8 or a list of ``[H,W,3]`` frames (frames of different sizes run image by image).
    ``compute`` / ``compute_one`` (tests) replace ``enhance_all(frames)`` / ``enhance(frame, strategy=k)``.
    """
    import time

    n = len(frames)
    filenames = list(filenames) if filenames is not None else [f"frame_{i:05d}" for i in range(n)]
    if len(filenames) != n:
        raise ValueError("one filename per frame")
    run_all = compute or (lambda f: enhance_all(f, device=device))
    run_one = compute_one or (lambda f, k: enhance(f, strategy=k, device=device))
    stats = {"total_images": n, "processed_images": 0, "failed_images": 0, "total_outputs": 0, "successful_outputs": 0,
             "failed_outputs": 0, "image_types": {"greenish": 0, "bluish": 0, "normal": 0}}
    names = [name for name, _ in DRIVER_STRATEGIES]
    outs = {name: [None] * n for name in names}
    # per image: (image_type, [(ok, time string or error message)] * 6) or None when the image failed before its strategy loop
    results = [None] * n

    def fused(idx, batch):
        t0 = time.time()
        o, types = run_all(batch)
        share = (time.time() - t0) / max(len(idx), 1)
        for j, i in enumerate(idx):
            for name in names:
                outs[name][i] = o[name][j]
            results[i] = (types[j], [(True, f"{share:.2f}s")] * len(names))

    def one_by_one(i):
        frame = frames[i]
        t0 = time.time()
        x = np.asarray(frame.cpu() if hasattr(frame, "cpu") else frame)
        try:
            # (the image_type column only; the strategies detect the cast again themselves, with the same result)
            kind = detect_image_type(x, device=device) if compute_one is None else "normal"
        except Exception:  # noqa: BLE001 - S6:508: the image fails as a whole
            return
        rows = []
        for k, name in enumerate(names, start=1):
            try:
                outs[name][i] = run_one(frame, k)
                rows.append((True, f"{time.time() - t0:.2f}s"))
            except Exception as e:  # noqa: BLE001 - S6:464: any failure of a strategy is a failed row
                outs[name][i] = None
                rows.append((False, str(e)[:50]))
        results[i] = (kind, rows)

    uniform = hasattr(frames, "shape") or len({tuple(np.shape(f)) for f in frames}) <= 1
    done = False
    if uniform and n > 0:
        try:
            batch = frames if hasattr(frames, "shape") else np.stack([np.asarray(f) for f in frames])
            fused(list(range(n)), batch)
            done = True
        except Exception:  # noqa: BLE001
            done = False
    if not done:
        for i in range(n):
            try:
                f = frames[i]
                fused([i], f[None] if hasattr(f, "shape") else np.asarray(f)[None])
            except Exception:  # noqa: BLE001
                one_by_one(i)

    rows = []
    for i, fname in enumerate(filenames):
        if results[i] is None:
            stats["failed_images"] += 1
            continue
        kind, per = results[i]
        stats["image_types"][kind] += 1
        good = 0
        for (sname, sdesc), (ok, info) in zip(DRIVER_STRATEGIES, per):
            if ok:
                rows.append({"filename": fname, "image_type": kind, "strategy": sname, "strategy_desc": sdesc,
                             "status": "success", "output_path": "", "processing_time": info})
                stats["successful_outputs"] += 1
                good += 1
            else:
                rows.append({"filename": fname, "image_type": kind, "strategy": sname, "strategy_desc": sdesc,
                             "status": "failed", "output_path": f"Error: {info}", "processing_time": "N/A"})
                stats["failed_outputs"] += 1
        stats["processed_images" if good else "failed_images"] += 1
        stats["total_outputs"] += len(DRIVER_STRATEGIES)
    return outs, rows, stats


# ------------------------------------------------------------------ vgg_16_UIE.DifferentiableEnhancement (N3)
class DifferentiableEnhancement:
    """Forward pass of ``vgg_16_UIE.DifferentiableEnhancement`` (vgg_16_UIE.py:24-128) on the device.

    ``forward(img, params)``: ``img`` is ``(B, 3, H, W)`` float32 (NumPy or torch ROCm tensor), ``params`` a dict of
    ``(B, 1)``-shaped values with the reference's keys: ``L_low`` and ``L_high`` are required, ``omega`` and ``gamma``
    optional (a missing key skips that stage, vgg_16_UIE.py:48,52).  Not differentiable: inference only.
    """

    device: int | None = None

    def forward(self, img, params):
        dev = get_device(self.device)
        was_numpy = not hasattr(img, "data_ptr")
        x = dev.tensor(np.ascontiguousarray(img, dtype=np.float32)) if was_numpy else img
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected a (B, 3, H, W) image batch, got {tuple(x.shape)}")
        B = x.shape[0]
        cols = []
        for key, default in (("L_low", None), ("L_high", None), ("omega", 0.0), ("gamma", 1.0)):
            if key in params:
                v = params[key]
                v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
                cols.append(np.broadcast_to(np.asarray(v, dtype=np.float32).reshape(-1), (B,)))
            elif default is None:
                raise KeyError(key)
            else:
                cols.append(np.full((B,), default, np.float32))
        pt = dev.tensor(np.ascontiguousarray(np.stack(cols, axis=1)))
        out = dev.diff_enhance_f32(x, pt, planar=True, has_omega="omega" in params, has_gamma="gamma" in params)
        return out.cpu().numpy() if was_numpy else out

    __call__ = forward

    def enhance_image(self, img, params):
        """``EnhancementPredictor.enhance_image(img, params)`` (use_trained_model.py:83-111) with explicit parameters:
        ``img`` HxWx3 RGB float in [0, 1], ``params`` a dict of Python floats with ``omega, gamma, L_low, L_high``."""
        dev = get_device(self.device)
        x = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
        if x.ndim != 3 or x.shape[2] != 3:
            raise ValueError(f"expected an HxWx3 image, got {x.shape}")
        pt = dev.tensor(np.array([[params["L_low"], params["L_high"], params["omega"], params["gamma"]]], np.float32))
        out = dev.diff_enhance_f32(dev.tensor(x[None]), pt, planar=False)[0].cpu().numpy()
        return np.clip(out, 0.0, 1.0)


QUALITY_KEYS = ("contrast", "sharpness", "entropy", "saturation", "brightness", "edge_density", "colorfulness", "naturalness")


class QualityAssessment:
    """Mirror of ``quality_assessment.QualityAssessment.comprehensive_assessment`` (quality_assessment.py:215-286)."""

    device: int | None = None

    @classmethod
    def comprehensive_assessment(cls, img, weights=None):
        """``img``: HxWx3 RGB float in [0, 1] -> ``(total_score, scores_dict)`` like the reference."""
        x = np.asarray(img)
        if x.ndim != 3 or x.shape[2] != 3:
            raise ValueError(f"expected an HxWx3 image, got {x.shape}")
        dev = get_device(cls.device)
        u8 = (x * 255).astype(np.uint8)  # quality_assessment.py:25 etc.: every score starts from this frame
        w = [weights.get(k, 0) for k in QUALITY_KEYS] if weights is not None else None
        row = dev.quality_scores(dev.tensor(u8[None]), dev.tensor(np.ascontiguousarray(x[None], dtype=np.float32)), w)
        row = row[0].cpu().numpy()
        return float(row[8]), {k: float(row[i]) for i, k in enumerate(QUALITY_KEYS)}


def quality_scores(frames_u8, frames_f32=None, weights=None, device: int | None = None):
    """Batch form on device or host arrays: ``[B,H,W,3]`` uint8 (and optionally the float32 images) -> ``[B,9]`` float64
    (eight scores in ``QUALITY_KEYS`` order, then the weighted total): best-of-N selection without leaving the GPU."""
    dev = get_device(device)
    batch, was_numpy, single = _as_batch_u8(frames_u8, dev)
    f32 = None
    if frames_f32 is not None:
        f32 = frames_f32 if hasattr(frames_f32, "data_ptr") else dev.tensor(np.ascontiguousarray(frames_f32, dtype=np.float32))
        if f32.dim() == 3:
            f32 = f32[None]
    w = [weights.get(k, 0) for k in QUALITY_KEYS] if isinstance(weights, dict) else weights
    return _finish(dev.quality_scores(batch, f32, w), was_numpy, single)


# config.py:28-75 (Config.STRATEGIES) and :77-84 (Config.QUALITY_WEIGHTS): the parameter sets and weights main.py labels with
CONFIG_STRATEGIES = {
    "strong_dehazing": {"name": "StrongDehazing", "omega": 0.5, "guided_radius": 15, "L_low": 10, "L_high": 95, "gamma": 1.2,
                        "apply_gamma": True},
    "medium_dehazing": {"name": "MediumDehazing", "omega": 0.6, "guided_radius": 20, "L_low": 15, "L_high": 92, "apply_gamma": True},
    "light_enhancement": {"name": "LightEnhancement", "omega": 0.4, "guided_radius": 10, "L_low": 15, "L_high": 95,
                          "apply_gamma": False},
    "clahe_enhancement": {"name": "CLAHEEnhancement", "clip_limit": 2.0, "tile_grid_size": (8, 8), "apply_gamma": False},
    "histogram_equalization": {"name": "HistogramEqualization", "L_low": 10, "L_high": 95},
}
CONFIG_QUALITY_WEIGHTS = {"contrast": 0.25, "sharpness": 0.20, "entropy": 0.15, "saturation": 0.15, "brightness": 0.15,
                          "edge_density": 0.10}


def _dict_params(dev, name, params):
    """uwie_params of apply_strategy(img, name, params) (ES:350-474: the keys a strategy reads, its in-code defaults otherwise)."""
    over = {}
    for key, field in (("omega", "omega"), ("guided_radius", "gf_ksize"), ("L_low", "L_low"), ("L_high", "L_high"),
                       ("clip_limit", "clip_limit"), ("gamma", "gamma")):
        if key in params:
            over[field] = params[key]
    if "tile_grid_size" in params:
        over["tiles_x"], over["tiles_y"] = (int(v) for v in params["tile_grid_size"])
    over["apply_gamma"] = int(bool(params.get("apply_gamma", False)))
    return dev.params(_lib.SURFACE_DICT, _lib.DICT_STRATEGIES[name], **over)


def select_best(frames, strategies=None, weights=None, device: int | None = None, return_all: bool = False):
    """The labelling loop of ``SelfSupervisedSystem.build_dataset`` (main.py:118-146) for one frame or a batch, in ONE device
    call: every entry of ``strategies`` (default ``Config.STRATEGIES``, config.py:28-75: ``{key: {'name': ..., params}}``) goes
    through ``apply_strategy``, ``comprehensive_assessment`` with ``weights`` (default ``Config.QUALITY_WEIGHTS``) scores each
    result, and the first maximum wins (main.py:145).  The three dehazing strategies share one atmospheric-light quadtree.

    ``frames``: uint8 ``[H,W,3]`` / ``[B,H,W,3]`` (main.py:108 makes ``u8.astype(float32) / 255`` of exactly these).
    Returns ``(best_names, best_images, scores)``: the winner's ``'name'`` per frame (a string for a single frame), its
    ``(enhanced * 255).astype(uint8)`` image (main.py:154-155 writes that), and ``scores[frame][name] = total``; with
    ``return_all`` a fourth value ``{name: uint8 batch}`` holds every strategy's output.  A strategy the device cannot
    run fails the call: fall back to ``EnhancementStrategies.apply_strategy`` + ``QualityAssessment`` per strategy, whose
    ``try`` blocks give a failing strategy the score 0.0 like main.py:139-142.
    """
    strategies = CONFIG_STRATEGIES if strategies is None else strategies
    weights = CONFIG_QUALITY_WEIGHTS if weights is None else weights
    dev = get_device(device)
    batch, was_numpy, single = _as_batch_u8(frames, dev)
    keys = list(strategies)
    for k in keys:
        if k not in _lib.DICT_STRATEGIES:
            raise ValueError(f"未知策略: {k}")
    plist = [_dict_params(dev, k, strategies[k]) for k in keys]
    names = [strategies[k].get("name", k) for k in keys]
    w = [weights.get(k, 0) for k in QUALITY_KEYS]
    best, img, scores, every = dev.select_best_u8(batch, plist, w, want_all=return_all)
    best_h = best.cpu().tolist()
    tot = scores[:, :, 8].cpu().numpy()
    dev.check_status()
    best_names = [names[i] for i in best_h]
    table = [{names[k]: float(tot[k, b]) for k in range(len(keys))} for b in range(len(best_h))]
    out = (best_names[0] if single else best_names, _finish(img, was_numpy, single), table[0] if single else table)
    if return_all:
        out = out + ({names[k]: _finish(every[k], was_numpy, single) for k in range(len(keys))},)
    return out


def extract_all_features(img, device: int | None = None):
    """``vgg_16_UIE.extract_all_features(img)`` (vgg_16_UIE.py:435-466) for a uint8 RGB frame ``[H,W,3]`` (returns the
    reference's ``(79,)`` float32 vector) or a batch ``[B,H,W,3]`` (returns ``(B, 79)``).  Float inputs of the reference
    (``_ensure_float01``) are not taken here: pass the decoded uint8 frame."""
    dev = get_device(device)
    if not hasattr(img, "data_ptr") and np.asarray(img).dtype != np.uint8:
        raise ValueError("extract_all_features on the device takes uint8 frames")
    batch, was_numpy, single = _as_batch_u8(img, dev)
    return _finish(dev.extract_features_u8(batch), was_numpy, single)


# ------------------------------------------------------------------ float <-> u8 bridging
def _recover_u8(img):
    """Invert ``u8.astype(float32)/255`` [+ ``color_correction``] exactly; returns (u8 frame, cast kind)."""
    x = np.asarray(img)
    if x.ndim != 3 or x.shape[2] != 3:
        raise ValueError(f"expected an HxWx3 image, got {x.shape}")
    if x.size == 0:
        raise ValueError("empty image")
    x32 = x.astype(np.float32)
    if x.dtype != np.float32 and not np.array_equal(x32.astype(x.dtype), x):
        raise UnsupportedInputError("float image is not float32-representable u8/255 data")
    table = np.arange(256, dtype=np.float32) / _F255
    for kind, chan in (("normal", None), ("greenish", 1), ("bluish", 2)):
        ok, u8 = True, np.empty(x.shape, np.uint8)
        for c in range(3):
            lut = table * _ATTEN if c == chan else table
            idx = np.clip(np.searchsorted(lut, x32[:, :, c]), 0, 255)
            if not np.array_equal(lut[idx], x32[:, :, c]):
                ok = False
                break
            u8[:, :, c] = idx
        if ok:
            return u8, kind
    raise UnsupportedInputError("float image is not u8-derived (u8/255, optionally colour-corrected): unsupported input")


def _float_kind(img):
    """'u8' for a u8 frame, 'derived' for a u8-derived float image, else the dtype name of a general float image."""
    x = np.asarray(img)
    if x.dtype == np.uint8:
        return "u8"
    try:
        _recover_u8(x)
        return "derived"
    except UnsupportedInputError:
        if x.dtype in (np.float32, np.float64):
            return x.dtype.name
        raise


def detect_image_type(img, device: int | None = None) -> str:
    """six_stadigy.py:292-302 on a float image (u8-derived: from its u8 frame; general float32: sequential mean on the
    device) or a u8 frame."""
    dev = get_device(device)
    what = _float_kind(img)
    if what == "float32":
        kind, _ = dev.cast_classify_f32(dev.tensor(np.ascontiguousarray(img)[None]))
        return _lib.CAST_KINDS[int(kind[0])]
    if what == "float64":
        raise UnsupportedInputError("detect_image_type: general float64 images are not supported (six_stadigy.py works on float32)")
    u8 = img if what == "u8" else _recover_u8(img)[0]
    kind, _ = dev.cast_classify(dev.tensor(u8[None]))
    return _lib.CAST_KINDS[int(kind[0])]


def color_correction(img, image_type: str, device: int | None = None):
    """six_stadigy.py:305-323.  ``"normal"`` returns the input object itself, like the reference."""
    if image_type not in ("greenish", "bluish"):
        return img
    dev = get_device(device)
    if _float_kind(img) == "float32":  # general float image
        k = torch.tensor([_lib.CAST_KINDS.index(image_type)], dtype=torch.int32, device=dev.torch_device)
        return dev.color_correct_f32(dev.tensor(np.ascontiguousarray(img)[None]), k)[0].cpu().numpy()
    u8, kind = _recover_u8(img)
    if kind != "normal":
        raise ValueError("image is already colour-corrected")
    k = torch.tensor([_lib.CAST_KINDS.index(image_type)], dtype=torch.int32, device=dev.torch_device)
    return dev.normalise_correct(dev.tensor(u8[None]), k)[0].cpu().numpy()


class SixStrategies:
    """Mirror of ``six_stadigy.EnhancementStrategies`` (strategy1..6: float32 HxWx3 in [0,1] -> float32 HxWx3)."""

    device: int | None = None

    @classmethod
    def _run(cls, number, img):
        dev = get_device(cls.device)
        what = _float_kind(img)
        if what == "float32":  # general float image: strategyN(img) as it stands (no cast detection inside, S6:230-285)
            p = dev.params(_lib.SURFACE_SIX, number, cast_correct=0, forced_cast=-1)
            return dev.enhance_float(dev.tensor(np.ascontiguousarray(img)[None]), p)[1][0].cpu().numpy()
        if what == "float64":
            raise UnsupportedInputError("six_stadigy strategies take float32 images (six_stadigy.py:406); a general float64 image "
                                        "is only supported on the dict surface (EnhancementStrategies.apply_strategy)")
        u8, kind = _recover_u8(img)
        # a colour-corrected input is replayed on the device from its u8 frame through a forced cast kind
        p = dev.params(_lib.SURFACE_SIX, number, cast_correct=0, forced_cast=_lib.CAST_KINDS.index(kind) or -1)
        _, outf = dev.enhance_u8(dev.tensor(u8[None]), p, want_float=True)
        return outf[0].cpu().numpy()

    @classmethod
    def strategy1_strong_dehazing(cls, img):
        return cls._run(1, img)

    @classmethod
    def strategy2_medium_dehazing(cls, img):
        return cls._run(2, img)

    @classmethod
    def strategy3_light_dehazing(cls, img):
        return cls._run(3, img)

    @classmethod
    def strategy4_clahe_enhancement(cls, img):
        return cls._run(4, img)

    @classmethod
    def strategy5_white_balance(cls, img):
        return cls._run(5, img)

    @classmethod
    def strategy6_histogram_eq(cls, img):
        return cls._run(6, img)


class EnhancementStrategies:
    """Mirror of ``enhancement_strategies.EnhancementStrategies.apply_strategy`` (ES:477-508)."""

    device: int | None = None
    swallow_errors = True  # ES:503-508 prints the failure and returns the input image

    @classmethod
    def apply_strategy(cls, img, strategy_name, params):
        if strategy_name not in _lib.DICT_STRATEGIES:
            raise ValueError(f"未知策略: {strategy_name}")
        try:
            return cls._run(img, strategy_name, params)
        except UnsupportedInputError:
            raise  # not a strategy failure: never handed back as if it had been processed
        except Exception as exc:  # noqa: BLE001 - mirrors the reference's blanket except
            if not cls.swallow_errors:
                raise
            print(f"策略 {strategy_name} 執行失敗: {exc}")
            return img

    @classmethod
    def _run(cls, img, name, params):
        dev = get_device(cls.device)
        x = np.asarray(img)
        if x.dtype not in (np.float32, np.float64):
            raise UnsupportedInputError(f"apply_strategy takes float32 / float64 images in [0, 1], got {x.dtype}")
        if x.ndim != 3 or x.shape[2] != 3 or x.size == 0:
            raise ValueError(f"expected a non-empty HxWx3 image, got {x.shape}")
        u8 = None
        if x.dtype == np.float32:  # main.py:108 hands over u8.astype(float32)/255: the fast path from the u8 frame
            try:
                cand, kind = _recover_u8(x)
                if kind == "normal":
                    u8 = cand
            except UnsupportedInputError:
                pass
        p = _dict_params(dev, name, params)
        # float64 like the reference (ES:247,307,345): a caller's (enhanced * 255).astype(np.uint8) (main.py:155) then
        # truncates the same values as with the reference
        if u8 is not None:
            return dev.enhance_u8_f64(dev.tensor(u8[None]), p)[1][0].cpu().numpy()
        # any other float32 / float64 image (the reference's harness inputs are np.random.rand): general float path
        return dev.enhance_float(dev.tensor(np.ascontiguousarray(x)[None]), p)[1][0].cpu().numpy()
