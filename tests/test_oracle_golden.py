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


# ------------------------------------------------------------------ vgg_16_UIE.DifferentiableEnhancement (N3)
def _vgg_cases():
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgg_stages.npz"))
    tags = sorted({k.split("/")[0] for k in z.files if not k.startswith("feat_")})
    return z, tags


def ulp_distance_f32(a, b):
    ia = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_diff_enhance_oracle_matches_the_reference_module():
    """oracle.diff_enhance against outputs of the real vgg_16_UIE.DifferentiableEnhancement (oracle/gen_golden_vgg.py).
    Stretch and dehazing are bit-exact; torch.pow on float32 may pick another CPU kernel on another machine: <= 1 ulp."""
    from oracle import uwie_oracle as orc

    z, tags = _vgg_cases()
    assert len(tags) == 5
    for tag in tags:
        par = {k: z[f"{tag}/{k}"] for k in ("L_low", "L_high", "omega", "gamma") if f"{tag}/{k}" in z.files}
        got = orc.diff_enhance(z[f"{tag}/img"], par)
        want = z[f"{tag}/out"]
        assert got.dtype == np.float32 and got.shape == want.shape
        if "gamma" in par:
            assert ulp_distance_f32(got, want).max() <= 1, tag
        else:
            assert np.array_equal(got, want), tag


def test_extract_all_features_oracle_matches_the_reference_function():
    from oracle import uwie_oracle as orc

    z, _ = _vgg_cases()
    tags = sorted({k.split("/")[0] for k in z.files if k.startswith("feat_")})
    assert len(tags) == 4
    for tag in tags:
        got = orc.extract_all_features(z[f"{tag}/u8"])
        assert got.dtype == np.float32 and got.shape == (79,)
        assert np.array_equal(got, z[f"{tag}/features"]), tag
