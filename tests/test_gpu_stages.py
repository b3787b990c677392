"""GPU parity, stage by stage: every HIP stage (through the C ABI) against the CPU oracle on the same seeded inputs.

Integer / byte stages and every float stage whose operation order is reproduced must match BIT FOR BIT; the
only tolerance is on ``gamma`` (device pow vs libm powf: <= 1 float32 ulp, stated in the test).
"""
import numpy as np
import pytest

from conftest import GOLDEN_TAGS

pytestmark = pytest.mark.gpu

KINDS = ["normal", "greenish", "bluish"]


@pytest.fixture(scope="module")
def dev():
    import underwater_image_enhancement_amd as uw

    return uw.get_device(0)


@pytest.fixture(scope="module")
def orc():
    from oracle import uwie_oracle

    return uwie_oracle


def frames_for_tests(rng):
    """name -> u8 frame.  Mixed content: noise, smooth casts, an odd size, tiny frames."""
    out = {}
    out["noise_61x83"] = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:120, 0:160]
    base = 0.5 + 0.2 * np.sin(xx / 17.0) * np.cos(yy / 13.0) + 0.15 * np.sin((xx + yy) / 29.0)
    for tag, gains in (("green_120x160", (0.45, 0.85, 0.80)), ("blue_120x160", (0.45, 0.75, 0.90))):
        f = base[:, :, None] * np.array(gains)[None, None, :] + rng.normal(0, 0.02, (120, 160, 3))
        out[tag] = np.clip(np.floor(255 * f), 0, 255).astype(np.uint8)
    out["hazy_97x131"] = np.floor(255 * (rng.random((97, 131, 3)) * 0.7 + 0.15)).astype(np.uint8)
    out["tiny_5x7"] = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    out["dark_40x50"] = rng.integers(0, 4, (40, 50, 3), dtype=np.uint8)
    return out


@pytest.fixture(scope="module")
def frames():
    return frames_for_tests(np.random.default_rng(4242))


def same(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype, (got.dtype, want.dtype)
    assert got.shape == want.shape, (got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.flatnonzero(got.ravel() != want.ravel())
        raise AssertionError(f"{bad.size} of {got.size} differ; first at {bad[0]}: got {got.ravel()[bad[0]]!r} "
                             f"want {want.ravel()[bad[0]]!r}")


# ------------------------------------------------------------------ entry stages
def test_cast_classify_matches_numpy_sequential_mean(dev, orc, frames, golden):
    rng = np.random.default_rng(1)
    cases = dict(frames)
    cases["noise_480x640"] = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    cases["bright_1080p"] = rng.integers(180, 256, (1080, 1920, 3), dtype=np.uint8)
    cases["sparse_300x400"] = ((rng.random((300, 400, 3)) < 0.002) * rng.integers(0, 256, (300, 400, 3))).astype(np.uint8)
    cases["zeros_64x64"] = np.zeros((64, 64, 3), np.uint8)
    cases["white_257x129"] = np.full((257, 129, 3), 255, np.uint8)  # (a ragged last chunk, every add of one size)
    cases["odd_1031x517"] = rng.integers(0, 256, (1031, 517, 3), dtype=np.uint8)
    cases["one_chunk_128x128"] = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)   # exactly 16384 pixels
    cases["chunk_plus_one_5x3277"] = rng.integers(0, 256, (5, 3277, 3), dtype=np.uint8)  # 16385: a one-pixel last chunk
    cases["bright_start_then_black"] = np.concatenate([np.full((40, 500, 3), 255, np.uint8), np.zeros((300, 500, 3), np.uint8)])
    cases["black_start_then_noise"] = np.concatenate([np.zeros((300, 500, 3), np.uint8), rng.integers(0, 256, (200, 500, 3), dtype=np.uint8)])
    cases["dark_4k"] = (rng.random((2160, 3840, 3)) < 0.3).astype(np.uint8) * rng.integers(0, 4, (2160, 3840, 3), dtype=np.uint8)
    cases["steps_2k"] = np.repeat(np.arange(0, 256, dtype=np.uint8), 3 * 2048 * 4).reshape(2048, 1024, 3)  # long constant runs
    for tag in GOLDEN_TAGS:
        cases["golden_" + tag] = golden[f"{tag}/u8"]
    for name, u8 in cases.items():
        x = orc.normalise_u8(u8)
        kind, mean = dev.cast_classify(dev.tensor(u8[None]))
        same(mean.cpu().numpy()[0], x.mean(axis=(0, 1)))
        assert KINDS[int(kind[0])] == orc.classify_cast(x), name
    for tag in GOLDEN_TAGS:  # the reference's own answers
        kind, mean = dev.cast_classify(dev.tensor(golden[f"{tag}/u8"][None]))
        same(mean.cpu().numpy()[0], golden[f"{tag}/cast_mean"])
        assert int(kind[0]) == int(golden[f"{tag}/cast_kind"])


def test_cast_classify_batch(dev, orc):
    rng = np.random.default_rng(2)
    batch = rng.integers(0, 256, (5, 70, 90, 3), dtype=np.uint8)
    batch[1, :, :, 1] = np.minimum(batch[1, :, :, 1].astype(int) + 60, 255)
    batch[3, :, :, 2] = np.minimum(batch[3, :, :, 2].astype(int) + 60, 255)
    kind, mean = dev.cast_classify(dev.tensor(batch))
    for b in range(5):
        x = orc.normalise_u8(batch[b])
        same(mean.cpu().numpy()[b], x.mean(axis=(0, 1)))
        assert KINDS[int(kind[b])] == orc.classify_cast(x)
    assert sorted(set(int(k) for k in kind)) == [0, 1, 2]


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_normalise_correct_matches_reference(dev, golden, tag):
    import torch

    u8 = golden[f"{tag}/u8"]
    for k, kind in enumerate(KINDS):
        kk = torch.tensor([k], dtype=torch.int32, device=dev.torch_device)
        same(dev.normalise_correct(dev.tensor(u8[None]), kk)[0].cpu().numpy(), golden[f"{tag}/corrected_{kind}"])


# ------------------------------------------------------------------ OpenCV primitives
def test_gray_lab_roundtrip_kernels(dev, orc):
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8)
    grid = np.stack(np.meshgrid(np.arange(0, 256, 3), np.arange(0, 256, 3), np.arange(0, 256, 5), indexing="ij"),
                    -1).reshape(1, -1, 86 * 52, 3).astype(np.uint8)
    for arr in (rgb, np.ascontiguousarray(grid)):
        t = dev.tensor(arr)
        for shift in (14, 15):
            same(dev.rgb2gray_u8(t, shift).cpu().numpy(),
                 np.stack([orc.cv_rgb2gray_u8(a, shift) for a in arr]))
        lab = dev.rgb2lab_u8(t)
        want_lab = np.stack([orc.cv_rgb2lab_u8(a) for a in arr])
        same(lab.cpu().numpy(), want_lab)
        same(dev.lab2rgb_u8(lab).cpu().numpy(), np.stack([orc.cv_lab2rgb_u8(a) for a in want_lab]))
    every_lab = rng.integers(0, 256, (1, 300, 300, 3), dtype=np.uint8)
    same(dev.lab2rgb_u8(dev.tensor(every_lab)).cpu().numpy(), orc.cv_lab2rgb_u8(every_lab[0])[None])


@pytest.mark.parametrize("shape,clip,tiles", [((64, 128), 2.0, (8, 8)), ((37, 53), 2.0, (8, 8)), ((120, 160), 4.0, (8, 8)),
                                              ((97, 131), 1.5, (8, 8)), ((48, 64), 3.0, (4, 6)), ((9, 11), 2.0, (8, 8)),
                                              ((270, 480), 40.0, (8, 8))])
def test_clahe_u8(dev, orc, shape, clip, tiles):
    rng = np.random.default_rng(5)
    planes = np.stack([rng.integers(0, 256, shape, dtype=np.uint8),
                       np.clip(rng.normal(120, 12, shape), 0, 255).astype(np.uint8),
                       np.full(shape, 77, np.uint8)])
    got = dev.clahe_u8(dev.tensor(planes), clip, tiles).cpu().numpy()
    same(got, np.stack([orc.cv_clahe_u8(p, clip, tiles) for p in planes]))


def test_equalize_hist_u8(dev, orc):
    rng = np.random.default_rng(6)
    planes = np.stack([rng.integers(0, 256, (50, 70), dtype=np.uint8), np.full((50, 70), 9, np.uint8),
                       (rng.random((50, 70)) < 0.5).astype(np.uint8) * 200,
                       np.clip(rng.normal(60, 5, (50, 70)), 0, 255).astype(np.uint8)])
    same(dev.equalize_hist_u8(dev.tensor(planes)).cpu().numpy(), np.stack([orc.cv_equalize_hist_u8(p) for p in planes]))


def test_canny_u8(dev, orc):
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:90, 0:120]
    smooth = (127 + 120 * np.sin(xx / 4.0) * np.cos(yy / 5.0)).astype(np.uint8)
    blobs = np.zeros((90, 120), np.uint8)
    blobs[20:60, 30:80] = 180
    blobs[35:50, 50:110] = 90
    spiral = np.zeros((90, 120), np.uint8)
    for t in np.linspace(0, 12 * np.pi, 4000):
        r = 3 + 1.1 * t
        y, x = int(45 + r * np.sin(t)), int(60 + r * np.cos(t))
        if 0 <= y < 90 and 0 <= x < 120:
            spiral[y, x] = min(255, int(30 + 6 * t))
    planes = np.stack([rng.integers(0, 256, (90, 120), dtype=np.uint8), smooth, blobs, spiral,
                       np.clip(smooth.astype(int) + rng.integers(-25, 25, smooth.shape), 0, 255).astype(np.uint8)])
    got = dev.canny_u8(dev.tensor(planes), 50, 150).cpu().numpy()
    want = np.stack([orc.cv_canny_u8(p, 50, 150) for p in planes])
    same(got, want)
    assert want[1].any() and want[2].any() and want[3].any()
    for shape in ((1, 9), (7, 1), (2, 2), (3, 5)):
        p = rng.integers(0, 256, (1,) + shape, dtype=np.uint8)
        same(dev.canny_u8(dev.tensor(p), 50, 150).cpu().numpy(), orc.cv_canny_u8(p[0], 50, 150)[None])


# ------------------------------------------------------------------ guided filter chain
@pytest.mark.parametrize("k", [15, 20, 10, 3, 1])
def test_box_filter_f64(dev, orc, k):
    rng = np.random.default_rng(8)
    planes = rng.random((3, 41, 67))
    same(dev.box_filter_f64(dev.tensor(planes), k).cpu().numpy(), np.stack([orc.cv_box_filter_f64(p, k) for p in planes]))
    small = rng.random((1, 6, 9))  # smaller than the window: multiple reflections
    same(dev.box_filter_f64(dev.tensor(small), k).cpu().numpy(), orc.cv_box_filter_f64(small[0], k)[None])


def test_transmission_init_and_guided_filter(dev, orc, frames):
    import torch

    S6 = orc.SixStrategyOracle
    rng = np.random.default_rng(9)
    for name, u8 in frames.items():
        x = orc.normalise_u8(u8)
        kind = orc.classify_cast(x)
        xc = orc.correct_cast(x, kind)
        A = (rng.random(3) * 0.6 + 0.3).astype(np.float32)
        kk = torch.tensor([KINDS.index(kind)], dtype=torch.int32, device=dev.torch_device)
        for omega, ks, eps in ((0.5, 15, 0.5), (0.3, 20, 0.5), (0.7, 10, 0.1)):
            p = dev.params(0, 2, omega=omega)
            t0, gray = dev.transmission_init(dev.tensor(u8[None]), dev.tensor(A[None]), kk, p)
            want_t0 = S6.transmission_init(xc, A, omega)
            want_gray = orc.cv_rgb2gray_u8((xc * 255).astype(np.uint8))
            same(t0[0].cpu().numpy(), want_t0)
            same(gray[0].cpu().numpy(), want_gray)
            want_t = S6.transmission(xc, A, omega, ks, eps)
            t = dev.guided_filter(gray, t0, ks, eps, exact=True)  # cv2.boxFilter's running-sum order, bit for bit
            same(t[0].cpu().numpy(), want_t)
            # default fused filter: same windows and borders, free float64 summation order.  Tolerance: 1e-11 absolute
            tf = dev.guided_filter(gray, t0, ks, eps, exact=False)[0].cpu().numpy()
            assert tf.dtype == np.float64 and np.abs(tf - want_t).max() <= 1e-11, (name, ks, np.abs(tf - want_t).max())
            # reduced-precision intermediates (uwie_params.inter_dtype = UWIE_INTER_FX32, BASELINE.json configs[4]): a and b
            # enter the ring as 32-bit fixed point.  Tolerance: 5e-10 absolute on t (k = 10 averages the fewest roundings)
            tx = dev.guided_filter(gray, t0, ks, eps, exact=2)[0].cpu().numpy()
            assert tx.dtype == np.float64 and np.abs(tx - want_t).max() <= 5e-10, (name, ks, np.abs(tx - want_t).max())


@pytest.mark.parametrize("k", [15, 20, 10])
@pytest.mark.parametrize("shape,bands", [((260, 700), 2), ((181, 256), 1), ((333, 490), 3), ((64, 300), 1), ((75, 128), 2)])
def test_split_ring_guided_filter(dev, orc, shape, bands, k):
    """The split-ring kernel (a in LDS, b in registers, steady loop unrolled over the ring period) for the reference's three
    window widths (six_stadigy.py:234,245,255; config.py:29-53).  k = 15: every row -- row indices are reflected at the top
    and bottom borders, a last period that runs past the image stores nothing there.  k = 20 / 10 (round 3): the window is
    not symmetric, so the split kernel takes the whole bands between row k and row H - (k - 2) and the general kernel the
    rows above and below; the raw rows of k = 20 are re-read instead of kept in registers.
    Production takes it for large batches only (4K x 16 and up: the full-size tests and bench.py); here the gf_bands
    selector forces it on small frames so that the oracle comparison covers every part of it: edge strips (reflected columns,
    mirrored a/b), interior strips, a ragged last strip, several bands (the last one shorter), heights that are no multiple
    of the ring period, frames barely taller than four windows (or shorter: then the general kernel alone runs).  Same
    tolerance as every float64 ring: 1e-11 absolute on t."""
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W + k)
    B = 2
    gray = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
    gray[1] = (np.add.outer(np.arange(H), 2 * np.arange(W)) % 256).astype(np.uint8)
    t0 = np.clip(rng.random((B, H, W), dtype=np.float32), 0.1, 1.0)
    with dev.tuning(gf_bands=bands):
        for eps in (0.5, 1e-3):
            got = dev.guided_filter(dev.tensor(gray), dev.tensor(t0), k, eps, exact=False).cpu().numpy()
            for b in range(B):
                want = np.clip(orc.guided_filter(gray[b].astype(np.float64) / 255.0, t0[b], k, eps), 0.1, 1.0)
                err = np.abs(got[b] - want).max()
                assert err <= 1e-11, (k, shape, bands, eps, b, err)
        with dev.tuning(gf_split=0):  # the general kernel alone gives the same answer to the same tolerance
            alone = dev.guided_filter(dev.tensor(gray), dev.tensor(t0), k, 0.5, exact=False).cpu().numpy()
        assert np.abs(alone - np.stack([np.clip(orc.guided_filter(gray[b].astype(np.float64) / 255.0, t0[b], k, 0.5), 0.1, 1.0)
                                        for b in range(B)])).max() <= 1e-11


@pytest.mark.parametrize("k", [15, 20, 10, 7])
@pytest.mark.parametrize("shape", [(83, 101), (130, 227), (97, 300), (64, 39), (211, 128)])
def test_fused_guided_filter_ragged_shapes(dev, orc, k, shape):
    """Default (gf_exact=0) guided filter on sizes that exercise every dispatch: one strip / several strips with a ragged
    last one, several bands, both image borders inside one strip, windows the wavefront kernel does not take (k=7) and
    frames too small for it (falls back to the LDS-tiled or the exact kernels).  Tolerance 1e-11 absolute on t."""
    H, W = shape
    rng = np.random.default_rng(100 * k + H)
    B = 3
    gray = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
    gray[1] = (np.add.outer(np.arange(H), np.arange(W)) % 256).astype(np.uint8)  # smooth ramp: tiny variances
    t0 = rng.random((B, H, W), dtype=np.float32)
    for eps in (0.5, 1e-3):
        got = dev.guided_filter(dev.tensor(gray), dev.tensor(t0), k, eps, exact=False).cpu().numpy()
        for b in range(B):
            want = np.clip(orc.guided_filter(gray[b].astype(np.float64) / 255.0, t0[b], k, eps), 0.1, 1.0)
            err = np.abs(got[b] - want).max()
            assert err <= 1e-11, (k, shape, eps, b, err)


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_restore_matches_reference(dev, golden, tag):
    import torch

    u8 = golden[f"{tag}/u8"]
    kk = torch.tensor([int(golden[f"{tag}/cast_kind"])], dtype=torch.int32, device=dev.torch_device)
    got = dev.restore(dev.tensor(u8[None]), dev.tensor(golden[f"{tag}/A"][None]), dev.tensor(golden[f"{tag}/t"][None]), kk)
    same(got[0].cpu().numpy(), golden[f"{tag}/s6_restore"])


def test_restore_shared_reciprocal_is_the_division(dev):
    """restore.h evaluates the three float64 quotients of a pixel with one reciprocal (the division's own Newton /
    Markstein sequence, shared).  Against NumPy's IEEE division on 6 M random (byte, A, t) triples: every float32 equal,
    t over the clip range [0.1, 1] with full mantissas, over ten decades, and outside the fast range (true division)."""
    import torch

    rng = np.random.default_rng(77)
    B, H, W = 4, 512, 1024
    u8 = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    A = rng.random((B, 3)).astype(np.float32)
    t = rng.uniform(0.1, 1.0, (B, H, W))
    t[1] = np.exp(rng.uniform(np.log(1e-5), np.log(1e5), (H, W)))
    t[2, ::2] = np.ldexp(rng.uniform(0.5, 1.0, t[2, ::2].shape), rng.integers(-140, -101, t[2, ::2].shape))
    t[2, 1::4] = np.ldexp(rng.uniform(0.5, 1.0, t[2, 1::4].shape), rng.integers(102, 140, t[2, 1::4].shape))
    t[3, :, ::3] = 1.0 - np.ldexp(1.0, -53)  # all-ones mantissa: the reciprocal's hard case
    kinds = np.array([0, 1, 2, 0], np.int32)
    got = dev.restore(dev.tensor(u8), dev.tensor(A), dev.tensor(t), torch.tensor(kinds, device=dev.torch_device)).cpu().numpy()
    x = u8.astype(np.float32) / np.float32(255.0)
    for b in range(B):
        if kinds[b]:
            x[b, :, :, kinds[b]] = x[b, :, :, kinds[b]] * np.float32(0.85)
    d = x - A[:, None, None, :]
    want = np.clip((d.astype(np.float64) / t[..., None] + A[:, None, None, :].astype(np.float64)).astype(np.float32), 0, 1)
    assert got.dtype == np.float32 and np.array_equal(got, want), int((got != want).sum())


def test_stretch_shared_reciprocal_is_the_division(dev):
    """The stretch divides every value of a plane by one denominator; the kernels keep the division's reciprocal and
    redo only its last five operations per value (devutil.h StretchDiv).  Against NumPy's IEEE float32 division on 3 M
    values per image: unit-range data, data with tiny and huge magnitudes (guarded: true division), a denominator near
    eps, and one with an all-ones mantissa."""
    rng = np.random.default_rng(78)
    B, H, W = 4, 500, 500
    img = rng.random((B, H, W, 3)).astype(np.float32)
    img[1] = (np.exp(rng.uniform(np.log(1e-30), np.log(1e30), (H, W, 3))) * rng.choice([-1, 1], (H, W, 3))).astype(np.float32)
    img[2] = np.float32(0.25) + (rng.random((H, W, 3)) * 3e-6).astype(np.float32)  # hi - lo of a few 1e-6
    img[3, ::2] = np.float32(1.0) - np.float32(2.0 ** -24)
    img[3, 1::2] = rng.random((H // 2, W, 3)).astype(np.float32) * np.float32(2.0 ** -24)
    got = dev.stretch_f32(dev.tensor(img), 1.0, 99.0).cpu().numpy()
    for b in range(B):
        for c in range(3):  # six_stadigy.py:191-199, as the oracle restates it
            plane = img[b, :, :, c]
            lo, hi = np.percentile(plane, 1.0), np.percentile(plane, 99.0)
            want = np.clip((plane - lo) / (hi - lo + 1e-6), 0, 1)
            assert want.dtype == np.float32 and np.array_equal(got[b, :, :, c], want), (b, c, int((got[b, :, :, c] != want).sum()))


# ------------------------------------------------------------------ percentiles / stretch / gamma / clahe on float images
@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_percentiles_and_stretch_match_reference(dev, golden, tag):
    img = golden[f"{tag}/s6_restore"]
    t = dev.tensor(img[None])
    qs = [5, 98, 15, 95]
    got = dev.percentiles_f32(t, qs).cpu().numpy()[0]
    want = np.array([[np.percentile(img[:, :, c], q) for q in qs] for c in range(3)], np.float32)
    same(got, want)
    for lo, hi in ((5, 98), (15, 95), (20, 85), (10, 95), (15, 90)):
        same(dev.stretch_f32(t, lo, hi)[0].cpu().numpy(), golden[f"{tag}/s6_contrast_{lo}_{hi}"])
    for p in (2, 3, 5):
        same(dev.stretch_f32(t, p, 100 - p)[0].cpu().numpy(), golden[f"{tag}/s6_wb_{p}"])


def test_percentiles_general_floats(dev):
    rng = np.random.default_rng(10)
    img = rng.normal(0, 3, (2, 83, 129, 3)).astype(np.float32)  # negative values, wide range
    img[1, :40] = 0.0  # heavy ties
    img[1, 40:, :, 2] = np.round(img[1, 40:, :, 2])  # quantised
    qs = [0, 2.5, 50, 99.9]
    got = dev.percentiles_f32(dev.tensor(img), qs).cpu().numpy()
    want = np.array([[[np.percentile(img[b, :, :, c], q) for q in qs] for c in range(3)] for b in range(2)], np.float32)
    same(got, want)


def test_fast_power_against_the_correctly_rounded_power(dev):
    """devutil.h pow_f32_fast (float64 exp2(g log2 x) with polynomial kernels good to ~2^-46): against the correctly
    rounded float32 power (float64 pow, rounded once) on 8 M arguments per exponent -- never more than one ulp away, and
    different at all for fewer than 1 in 10^5 arguments; zero, tiny and subnormal results included."""
    rng = np.random.default_rng(99)
    n = 1 << 23
    x = rng.random(n).astype(np.float32)
    x[: n // 4] = np.exp(rng.uniform(np.log(1e-38), 0.0, n // 4)).astype(np.float32)  # all magnitudes
    x[0], x[1], x[2] = 0.0, 1.0, np.float32(1e-8)
    xt = dev.tensor(x.reshape(1, 2048, -1, 1).repeat(3, axis=3))
    for g in (0.3, 1 / 2.2, 0.5, 1.2, 1.5, 2.2, 3.0, 17.3):
        got = dev.gamma_f32(xt, g, 1).cpu().numpy()[..., 0].reshape(-1)
        with np.errstate(under="ignore"):
            want = np.power(x.astype(np.float64), np.float64(np.float32(g))).astype(np.float32)
        d = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert d.max() <= 1, (g, int(d.max()))
        assert np.count_nonzero(d) <= n // 100000, (g, int(np.count_nonzero(d)))


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_gamma_within_one_ulp_of_reference(dev, golden, tag):
    img = golden[f"{tag}/s6_restore"]
    for g in (1.5, 1.3, 1.2, 1.4):
        got = dev.gamma_f32(dev.tensor(img[None]), g, 1)[0].cpu().numpy()
        want = golden[f"{tag}/s6_gamma_{g}"]
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        # Tolerance: 1 float32 ulp.  NumPy's float32 power is a SIMD routine that is itself only faithfully
        # rounded (and CPU-dispatch dependent), so bit equality with it is not defined across machines; the device
        # value is the correctly rounded one, which is checked against a float64 evaluation below.
        assert ulp.max() <= 1, f"gamma {g}: {ulp.max()} ulp"
        exact = np.power(img.astype(np.float64), np.float64(np.float32(g))).astype(np.float32)
        assert (got != exact).mean() < 1e-5
        # after the output quantisation (x*255 -> u8, six_stadigy.py:430) the results are identical or 1 LSB apart
        q = np.abs((got * 255).astype(np.uint8).astype(int) - (want * 255).astype(np.uint8).astype(int))
        assert q.max() <= 1


def test_clahe_f32(dev, orc, frames):
    S6 = orc.SixStrategyOracle
    for name, u8 in frames.items():
        x = orc.normalise_u8(u8)
        for clip in (2.0, 4.0):
            same(dev.clahe_f32(dev.tensor(x[None]), clip)[0].cpu().numpy(), S6.clahe(x, clip))


# ------------------------------------------------------------------ atmospheric light
def test_atmospheric_light_matches_oracle_trace(dev, orc, frames):
    import torch

    rng = np.random.default_rng(11)
    cases = dict(frames)
    cases["noise_200x333"] = rng.integers(0, 256, (200, 333, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:256, 0:320]
    f = (0.3 + 0.5 * np.exp(-((xx - 230) ** 2 + (yy - 60) ** 2) / 4000.0))[:, :, None] * np.array([0.4, 0.9, 1.0])
    cases["glow_256x320"] = np.clip(255 * (f + rng.normal(0, 0.01, f.shape)), 0, 255).astype(np.uint8)
    for name, u8 in cases.items():
        x = orc.normalise_u8(u8)
        kind = orc.classify_cast(x)
        xc = orc.correct_cast(x, kind)
        trace = []
        want_A = orc.atmospheric_light(xc, 1, trace=trace)
        kk = torch.tensor([KINDS.index(kind)], dtype=torch.int32, device=dev.torch_device)
        A, tr = dev.atmospheric_light(dev.tensor(u8[None]), kk, trace=True)
        for lvl, (y0, x0, rows, cols, scores) in enumerate(trace):
            rec = tr[0, lvl]
            assert (rec["y0"], rec["x0"], rec["rows"], rec["cols"]) == (y0, x0, rows, cols), (name, lvl)
            same(rec["score"], np.array(scores, np.float64))
        assert tr[0, len(trace)]["rows"] == 0
        same(A[0].cpu().numpy(), np.asarray(want_A))


def test_quadtree_levels_decided_from_histograms_equal_the_exact_kernels(dev, orc):
    """Round 3: a launched quadtree level is decided from byte histograms of the quadrants when the score intervals
    separate (k_q_hist / k_q_decide), by the exact NumPy-order kernels otherwise (tuning q_hist = 2 forces that, 0 switches
    the histograms off; a trace request takes the exact kernels too).  All routes must return the same atmospheric light --
    smooth frames (decided), uniform noise (near ties), flat and two-level frames (exact ties: first maximum wins), a frame
    whose bright quadrant is the last one, every cast kind, a batch mixing them -- and the oracle's."""
    import torch

    rng = np.random.default_rng(2718)
    H, W = 460, 700
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    f = (0.3 + 0.5 * np.exp(-((xx - 520) ** 2 + (yy - 330) ** 2) / 30000.0))[:, :, None] * np.array([0.45, 0.85, 0.8])
    frames.append(np.clip(255 * (f + rng.normal(0, 0.02, f.shape)), 0, 255).astype(np.uint8))
    frames.append(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
    frames.append(np.full((H, W, 3), 97, np.uint8))
    two = np.full((H, W, 3), 60, np.uint8)
    two[H // 2:, W // 2:] = 200
    frames.append(two)
    frames.append(np.clip(rng.normal(128, 3, (H, W, 3)), 0, 255).astype(np.uint8))
    frames.append(np.zeros((H, W, 3), np.uint8))
    batch = np.stack(frames)
    for kind_id in (0, 1, 2):
        kk = torch.full((len(frames),), kind_id, dtype=torch.int32, device=dev.torch_device)
        got = {}
        for q_hist in (1, 0, 2):
            with dev.tuning(q_hist=q_hist):
                got[q_hist] = dev.atmospheric_light(dev.tensor(batch), kk).cpu().numpy()
        assert np.array_equal(got[1], got[0]) and np.array_equal(got[1], got[2]), kind_id
        for i, u8 in enumerate(frames):
            xc = orc.correct_cast(orc.normalise_u8(u8), KINDS[kind_id])
            same(got[1][i], np.asarray(orc.atmospheric_light(xc, 1)))
    # one large frame: four launched levels at 1500 x 2100, every one of them through both routes
    big = np.clip(255 * ((0.35 + 0.4 * np.sin(np.mgrid[0:1500, 0:2100][1] / 300.0) * np.cos(np.mgrid[0:1500, 0:2100][0] / 170.0))[:, :, None]
                         * np.array([0.5, 0.9, 0.8]) + rng.normal(0, 0.02, (1500, 2100, 3))), 0, 255).astype(np.uint8)
    kk = torch.zeros(1, dtype=torch.int32, device=dev.torch_device)
    res = []
    for q_hist in (1, 0, 2):
        with dev.tuning(q_hist=q_hist):
            res.append(dev.atmospheric_light(dev.tensor(big[None]), kk).cpu().numpy())
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])


def test_atmospheric_light_other_leaf_sizes(dev, orc):
    """min_size > 1 (the quadtree's leaf size, S6:49): the walk stops earlier -- inside the launched levels, at the hand-over
    to k_q_tail, or inside it -- and the brightest pixel is taken over a larger leaf."""
    import torch

    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:300, 0:420]
    f = (0.35 + 0.4 * np.exp(-((xx - 300) ** 2 + (yy - 90) ** 2) / 9000.0))[:, :, None] * np.array([0.5, 0.85, 0.9])
    u8 = np.clip(255 * (f + rng.normal(0, 0.02, f.shape)), 0, 255).astype(np.uint8)
    x = orc.normalise_u8(u8)
    kind = orc.classify_cast(x)
    xc = orc.correct_cast(x, kind)
    kk = torch.tensor([KINDS.index(kind)], dtype=torch.int32, device=dev.torch_device)
    for min_size in (2, 9, 40, 75, 150, 400):
        p = dev.params(0, 2, min_size=min_size)
        trace = []
        want_A = orc.atmospheric_light(xc, min_size, trace=trace)
        A, tr = dev.atmospheric_light(dev.tensor(u8[None]), kk, p=p, trace=True)
        assert tr[0, len(trace)]["rows"] == 0, min_size
        for lvl, (y0, x0, rows, cols, scores) in enumerate(trace):
            rec = tr[0, lvl]
            assert (rec["y0"], rec["x0"], rec["rows"], rec["cols"]) == (y0, x0, rows, cols), (min_size, lvl)
            same(rec["score"], np.array(scores, np.float64))
        same(A[0].cpu().numpy(), np.asarray(want_A))


def test_quadtree_edge_counts_with_isolated_strong_pixels(dev, orc):
    """The Canny pre-pass (k_canny_strong: 256-column strips, 32-row bands, region borders replicated) and k_q_tail's
    in-LDS Canny may never lose an edge: flat frames with one bright pixel at strip / band / quadrant borders, compared
    with the oracle's trace level by level (the edge term is the only thing that separates the quadrants' scores)."""
    import torch

    H, W = 300, 1100
    spots = [(0, 0), (149, 549), (150, 550), (31, 255), (32, 256), (33, 257), (299, 1099), (75, 1098), (64, 511), (150, 0),
             (0, 550), (299, 549)]
    for i, (py, px) in enumerate(spots):
        u8 = np.full((H, W, 3), 90, np.uint8)
        u8[..., 1] = 120
        u8[py, px] = 255
        if i % 3 == 2:  # a second, weak-only neighbourhood elsewhere: must not count
            u8[(py + 97) % H, (px + 301) % W] = 112
        x = orc.normalise_u8(u8)
        kind = orc.classify_cast(x)
        xc = orc.correct_cast(x, kind)
        trace = []
        want_A = orc.atmospheric_light(xc, 1, trace=trace)
        kk = torch.tensor([KINDS.index(kind)], dtype=torch.int32, device=dev.torch_device)
        A, tr = dev.atmospheric_light(dev.tensor(u8[None]), kk, trace=True)
        for lvl, (y0, x0, rows, cols, scores) in enumerate(trace):
            rec = tr[0, lvl]
            assert (rec["y0"], rec["x0"], rec["rows"], rec["cols"]) == (y0, x0, rows, cols), ((py, px), lvl)
            same(rec["score"], np.array(scores, np.float64))
        same(A[0].cpu().numpy(), np.asarray(want_A))


# ------------------------------------------------------------------ vgg_16_UIE.DifferentiableEnhancement (N3)
def test_diff_enhance_matches_reference_outputs_and_oracle(dev, orc):
    """uwie_diff_enhance_f32 against the real module's outputs (tests/golden/vgg_stages.npz) and the torch-CPU oracle:
    bit-exact without gamma; with gamma <= 1 float32 ulp (the device rounds pow once from float64, torch's float32
    pow is a 1-ulp routine) -- except where the clamp at 1.0 hides the difference."""
    import os

    import torch

    import underwater_image_enhancement_amd as uw
    from test_oracle_golden import ulp_distance_f32

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgg_stages.npz"))
    tags = sorted({k.split("/")[0] for k in z.files if not k.startswith("feat_")})
    enh = uw.DifferentiableEnhancement()
    for tag in tags:
        par = {k: z[f"{tag}/{k}"] for k in ("L_low", "L_high", "omega", "gamma") if f"{tag}/{k}" in z.files}
        got = enh(z[f"{tag}/img"], par)
        want = z[f"{tag}/out"]
        assert got.dtype == np.float32 and got.shape == want.shape
        if "gamma" in par:
            assert ulp_distance_f32(got, want).max() <= 1, tag
        else:
            same(got, want)
    # larger seeded cases against the oracle, both layouts, torch tensors in and out
    rng = np.random.default_rng(31)
    img = rng.random((2, 3, 211, 157), dtype=np.float32)
    par = {"L_low": np.array([[7.5], [22.0]], np.float32), "L_high": np.array([[91.0], [70.0]], np.float32),
           "omega": np.array([[0.35], [0.8]], np.float32), "gamma": np.array([[0.6], [2.4]], np.float32)}
    want = orc.diff_enhance(img, par)
    got = enh(dev.tensor(img), {k: torch.from_numpy(v) for k, v in par.items()})
    assert isinstance(got, torch.Tensor) and got.is_cuda
    assert ulp_distance_f32(got.cpu().numpy(), want).max() <= 1
    nog = {k: v for k, v in par.items() if k != "gamma"}
    same(enh(img, nog), orc.diff_enhance(img, nog))
    hwc = np.ascontiguousarray(img[0].transpose(1, 2, 0))
    p1 = {"L_low": 7.5, "L_high": 91.0, "omega": 0.35, "gamma": 0.6}
    assert ulp_distance_f32(enh.enhance_image(hwc, p1), orc.diff_enhance_image(hwc, p1)).max() <= 1


# ------------------------------------------------------------------ vgg_16_UIE.extract_all_features (N4)
def test_extract_all_features_bit_exact(dev, orc):
    """uwie_extract_features_u8 against the reference function's outputs (goldens) and the NumPy oracle on further
    sizes: ragged buffers, a single full 8192 buffer, odd and even pixel counts, a batch.  Bit-exact."""
    import os

    import underwater_image_enhancement_amd as uw

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgg_stages.npz"))
    for tag in sorted({k.split("/")[0] for k in z.files if k.startswith("feat_")}):
        got = uw.extract_all_features(z[f"{tag}/u8"])
        assert got.shape == (79,) and got.dtype == np.float32
        same(got, z[f"{tag}/features"])
    rng = np.random.default_rng(64)
    for shape in ((64, 128), (211, 157), (480, 640), (33, 83), (128, 64)):
        u8 = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        u8[: shape[0] // 3] //= 3
        same(uw.extract_all_features(u8), orc.extract_all_features(u8))
    batch = rng.integers(0, 256, (3, 70, 90, 3), dtype=np.uint8)
    got = uw.extract_all_features(batch)
    assert got.shape == (3, 79)
    for b in range(3):
        same(got[b], orc.extract_all_features(batch[b]))


# ------------------------------------------------------------------ quality_assessment.QualityAssessment (N2)
def test_quality_scores_match_oracle(dev, orc):
    """uwie_quality_scores against the oracle's restatement of quality_assessment.py.  edge_density and naturalness are
    integer-derived (exact up to float64 rounding); the others are float32 NumPy statistics that the device evaluates in
    float64: tolerance 2e-3 on the 0..100 scale."""
    import underwater_image_enhancement_amd as uw

    rng = np.random.default_rng(2025)
    yy, xx = np.mgrid[0:96, 0:130]
    smooth = 0.5 + 0.3 * np.sin(xx / 17.0) * np.cos(yy / 11.0)
    images = {
        "random": rng.random((96, 130, 3)),
        "dark": rng.random((96, 130, 3)) * 0.3,
        "bright": 0.7 + rng.random((96, 130, 3)) * 0.3,
        "flat": np.full((64, 64, 3), 0.5),
        "binary": rng.choice([0.0, 1.0], size=(70, 90, 3)),
        "smooth": np.clip(smooth[:, :, None] * np.array([0.5, 0.8, 0.9]) + rng.normal(0, 0.01, (96, 130, 3)), 0, 1),
    }
    for name, img in images.items():
        img = img.astype(np.float32)
        total, scores = uw.QualityAssessment.comprehensive_assessment(img)
        wtotal, want = orc.quality_assessment(img)
        assert list(scores) == list(uw.QUALITY_KEYS)
        for k in uw.QUALITY_KEYS:
            tol = 1e-9 if k in ("edge_density", "naturalness") else 2e-3
            assert abs(scores[k] - float(want[k])) <= tol, (name, k, scores[k], float(want[k]))
        assert abs(total - float(wtotal)) <= 2e-3, (name, total, float(wtotal))
    # batch form on the enhancement outputs, custom weights
    u8 = (images["smooth"].astype(np.float32) * 255).astype(np.uint8)
    batch = np.stack([u8, u8[::-1].copy()])
    w = {"contrast": 0.5, "entropy": 0.5}
    got = uw.quality_scores(batch, weights=w)
    assert got.shape == (2, 9)
    for b in range(2):
        wt, sc = orc.quality_assessment(batch[b].astype(np.float32) / 255.0, weights=w)
        assert abs(got[b, 8] - float(wt)) <= 2e-3
