"""Generate tests/golden/numpy_stages.npz from the REAL reference (this container only).

TEST INFRASTRUCTURE.  The reference (/root/reference, pure Python) is imported
with inert stand-ins for the modules that are not installed here (cv2, skimage)
-- any attribute access on a stand-in raises, so only the reference's NumPy-only
functions can execute.  Their inputs (seeded) and outputs are stored as small
fixtures; tests/test_oracle_golden.py replays them against oracle/uwie_oracle.py
and the GPU tests replay them against the HIP path.  Nothing from the reference
travels: the fixture holds arrays only.

Functions captured (all NumPy-only in the reference):
  six_stadigy.py: restore_image :183, enhance_contrast :191, white_balance :211,
                  gamma_correction :222, get_brightest_pixel :160,
                  detect_image_type :292, color_correction :305
  enhancement_strategies.py: recover_image :237, color_enhancement :252,
                  gamma_correction :276, get_brightest_pixel :191

Run:  python oracle/gen_golden.py   (NumPy 2.2.6; results of np.percentile are
version dependent -- see SURVEY.md section 8c).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "numpy_stages.npz")


class _InertAttr:
    """Placeholder for a missing library symbol: importing it works, using it raises."""

    def __init__(self, qualname):
        self._qualname = qualname

    def _refuse(self, *args, **kwargs):
        raise RuntimeError(f"{self._qualname} is not available in this container")

    __call__ = __getattr__ = _refuse


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertAttr(f"{self.__name__}.{name}")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("cv2", "skimage", "skimage.measure", "skimage.color", "skimage.exposure", "skimage.feature"):
        mod = _Inert(name)
        mod.__path__ = []  # behave like a package for ``from skimage import x``
        sys.modules.setdefault(name, mod)
    sk = sys.modules["skimage"]
    for sub in ("measure", "color", "exposure", "feature"):
        sk.__dict__[sub] = sys.modules["skimage." + sub]
    sys.path.insert(0, REF)
    import enhancement_strategies as es  # noqa: E402
    import six_stadigy as s6  # noqa: E402
    sys.path.remove(REF)
    return s6, es


def frames(rng):
    """Seeded u8 frames: uniform noise, a hazy band, greenish and bluish casts, one odd size."""
    out = {}
    out["uniform_48x64"] = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    out["hazy_37x53"] = np.floor(255 * (rng.random((37, 53, 3)) * 0.7 + 0.15)).astype(np.uint8)
    yy, xx = np.mgrid[0:40, 0:56]
    base = 0.55 + 0.25 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    for tag, gains in (("greenish_40x56", (0.45, 0.85, 0.80)), ("bluish_40x56", (0.45, 0.75, 0.90))):
        f = base[:, :, None] * np.array(gains)[None, None, :] + rng.normal(0, 0.02, (40, 56, 3))
        out[tag] = np.clip(np.floor(255 * f), 0, 255).astype(np.uint8)
    return out


def main():
    s6, es = import_reference()
    S6, ES = s6.EnhancementStrategies, es.EnhancementStrategies
    rng = np.random.default_rng(20251121)
    gold = {}
    for tag, u8 in frames(rng).items():
        x = u8.astype(np.float32) / 255.0  # six_stadigy.py:406
        gold[f"{tag}/u8"] = u8
        kind = s6.detect_image_type(x)
        gold[f"{tag}/cast_kind"] = np.array(["normal", "greenish", "bluish"].index(kind), np.int32)
        gold[f"{tag}/cast_mean"] = x.mean(axis=(0, 1))
        for forced in ("greenish", "bluish", "normal"):
            gold[f"{tag}/corrected_{forced}"] = s6.color_correction(x, forced)
        xc = s6.color_correction(x, kind)
        A = rng.random(3).astype(np.float32) * np.float32(0.5) + np.float32(0.5)
        t = rng.random(x.shape[:2]) * 0.9 + 0.1  # float64 transmission in [0.1, 1]
        gold[f"{tag}/A"] = A
        gold[f"{tag}/t"] = t
        restored = S6.restore_image(xc, A, t)
        gold[f"{tag}/s6_restore"] = restored
        for lo, hi in ((5, 98), (15, 95), (20, 85), (10, 95), (15, 90)):
            gold[f"{tag}/s6_contrast_{lo}_{hi}"] = S6.enhance_contrast(restored, lo, hi)
        for p in (2, 3, 5):
            gold[f"{tag}/s6_wb_{p}"] = S6.white_balance(restored, p)
        for g in (1.5, 1.3, 1.2, 1.4):
            gold[f"{tag}/s6_gamma_{g}"] = S6.gamma_correction(restored, g)
        gold[f"{tag}/s6_brightest"] = np.array(S6.get_brightest_pixel(xc))
        gold[f"{tag}/s6_brightest_sub"] = np.array(S6.get_brightest_pixel(xc[3:4, 5:8, :]))
        # enhancement_strategies.py surface (A tiled to HxWx3, float32 in / float64 out)
        At = np.tile(A.reshape(1, 1, 3), (x.shape[0], x.shape[1], 1))
        rec = ES.recover_image(x, t, At)
        gold[f"{tag}/es_recover"] = rec
        for lo, hi in ((10, 95), (15, 92), (15, 95), (20, 85)):
            gold[f"{tag}/es_stretch_{lo}_{hi}"] = ES.color_enhancement(rec, lo, hi)
        gold[f"{tag}/es_stretch_f32_15_95"] = ES.color_enhancement(x, 15, 95)
        gold[f"{tag}/es_gamma_1.2"] = ES.gamma_correction(rec, 1.2)
        gold[f"{tag}/es_brightest"] = np.array(ES.get_brightest_pixel(x))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **gold)
    print(f"wrote {OUT}: {len(gold)} arrays, numpy {np.__version__}")


if __name__ == "__main__":
    main()
