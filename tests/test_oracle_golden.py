"""The CPU oracle against golden vectors captured from the real reference.

The fixture tests/golden/numpy_stages.npz was produced by oracle/gen_golden.py,
which imports /root/reference with cv2/skimage stubbed and runs the reference's
NumPy-only functions on seeded inputs.  Everything here must match BIT FOR BIT.
"""
import numpy as np
import pytest

from conftest import GOLDEN_TAGS
from oracle import uwie_oracle as orc

S6 = orc.SixStrategyOracle
ES = orc.DictStrategyOracle
KINDS = ["normal", "greenish", "bluish"]


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert a.shape == b.shape
    assert np.array_equal(a, b)


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_cast_detection_and_correction(golden, tag):
    x = orc.normalise_u8(golden[f"{tag}/u8"])
    same(x.mean(axis=(0, 1)), golden[f"{tag}/cast_mean"])
    assert KINDS.index(orc.classify_cast(x)) == int(golden[f"{tag}/cast_kind"])
    for kind in KINDS:
        same(orc.correct_cast(x, kind), golden[f"{tag}/corrected_{kind}"])
    assert orc.correct_cast(x, "normal") is x  # six_stadigy.py:323 returns the same object


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_six_strategy_numpy_stages(golden, tag):
    x = orc.normalise_u8(golden[f"{tag}/u8"])
    xc = orc.correct_cast(x, KINDS[int(golden[f"{tag}/cast_kind"])])
    restored = S6.restore(xc, golden[f"{tag}/A"], golden[f"{tag}/t"])
    same(restored, golden[f"{tag}/s6_restore"])
    for lo, hi in ((5, 98), (15, 95), (20, 85), (10, 95), (15, 90)):
        same(S6.stretch(restored, lo, hi), golden[f"{tag}/s6_contrast_{lo}_{hi}"])
    for p in (2, 3, 5):
        same(S6.white_balance(restored, p), golden[f"{tag}/s6_wb_{p}"])
    for g in (1.5, 1.3, 1.2, 1.4):
        same(S6.gamma(restored, g), golden[f"{tag}/s6_gamma_{g}"])
    same(orc.brightest_pixel(xc), golden[f"{tag}/s6_brightest"])
    same(orc.brightest_pixel(xc[3:4, 5:8, :]), golden[f"{tag}/s6_brightest_sub"])


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_dict_strategy_numpy_stages(golden, tag):
    x = orc.normalise_u8(golden[f"{tag}/u8"])
    A = golden[f"{tag}/A"]
    At = np.tile(A.reshape(1, 1, 3), (x.shape[0], x.shape[1], 1))
    rec = ES.recover(x, golden[f"{tag}/t"], At)
    same(rec, golden[f"{tag}/es_recover"])
    for lo, hi in ((10, 95), (15, 92), (15, 95), (20, 85)):
        same(ES.stretch(rec, lo, hi), golden[f"{tag}/es_stretch_{lo}_{hi}"])
    same(ES.stretch(x, 15, 95), golden[f"{tag}/es_stretch_f32_15_95"])
    same(ES.gamma(rec, 1.2), golden[f"{tag}/es_gamma_1.2"])
    same(orc.brightest_pixel(x), golden[f"{tag}/es_brightest"])
