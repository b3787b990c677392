"""GPU parity of the whole pipeline: uwie_enhance_u8 (all six six_stadigy.py strategies) against the CPU oracle.

The bar is <= 1 LSB on uint8 (BASELINE.json); the tests additionally report how many pixels differ at all.
"""
import numpy as np
import pytest

from test_gpu_stages import frames_for_tests

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uw():
    import underwater_image_enhancement_amd as uw

    return uw


@pytest.fixture(scope="module")
def orc():
    from oracle import uwie_oracle

    return uwie_oracle


def check_u8(got, want, what):
    assert got.shape == want.shape and got.dtype == np.uint8
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, f"{what}: max |delta| = {d.max()} LSB, {np.count_nonzero(d > 1)} pixels beyond 1 LSB"
    return int(np.count_nonzero(d))


@pytest.mark.parametrize("strategy", [1, 2, 3, 4, 5, 6])
def test_enhance_matches_oracle(uw, orc, strategy):
    frames = frames_for_tests(np.random.default_rng(4242))
    frames.pop("tiny_5x7")
    total = diff = 0
    for name, u8 in frames.items():
        got = uw.enhance(u8, strategy=strategy)
        want = orc.enhance_u8(u8, strategy)
        diff += check_u8(got, want, f"strategy {strategy} on {name}")
        total += got.size
    assert diff == 0, f"strategy {strategy}: {diff} of {total} bytes differ by 1 LSB"


@pytest.mark.parametrize("shape", [(5, 7), (9, 300), (300, 9), (2, 3), (1, 1), (1, 40), (16, 16)])
def test_frames_smaller_than_the_filter_window_and_the_tile_grid(uw, orc, shape):
    """frames narrower than the guided filter's window (10 / 15 / 20), than CLAHE's 8 x 8 tile grid and
    than one quadtree split, end to end on both surfaces against the oracle (which follows the reference: cv2.boxFilter's
    BORDER_REFLECT_101 of a source shorter than the window, CLAHE's reflected padding up to the tile grid, a quadtree that
    stops at once).  Bit-identical, gamma strategies within 1 LSB."""
    rng = np.random.default_rng(1000 + shape[0] * 31 + shape[1])
    u8 = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    for k in range(1, 7):
        n = check_u8(uw.enhance(u8, strategy=k), orc.enhance_u8(u8, k), f"strategy {k} on {shape}")
        assert n == 0 or k in (1, 4, 5, 6), f"strategy {k} on {shape}: {n} bytes differ"
    x = orc.normalise_u8(u8)
    for name in ("strong_dehazing", "medium_dehazing", "light_enhancement", "clahe_enhancement", "histogram_equalization"):
        params = orc.CONFIG_STRATEGIES[name]
        want = (orc.DictStrategyOracle.run(x, name, params) * 255).astype(np.uint8)
        got = (uw.EnhancementStrategies.apply_strategy(x, name, params) * 255).astype(np.uint8)
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1, f"{name} on {shape}: {d.max()} LSB"


def test_enhance_ragged_midsize(uw, orc):
    """Frames large enough for the launched quadtree levels (level 0 writes the gray plane on the way: GrayOut in
    k_airlight.hip) with sizes that put leaves across row ends and leave ragged chunks: the gray plane feeds the guided
    filter everywhere, so the whole output checks it."""
    rng = np.random.default_rng(77)
    for H, W in ((333, 517), (203, 1001)):
        yy, xx = np.mgrid[0:H, 0:W]
        field = 0.5 + 0.3 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
        f = field[:, :, None] * np.array([0.5, 0.8, 0.9]) + rng.normal(0, 0.03, (H, W, 3))
        u8 = np.clip(np.floor(255 * f), 0, 255).astype(np.uint8)
        u8[H // 3: H // 3 + 40, W // 4: W // 4 + 60] = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)  # a textured patch
        for strategy in (2, 3):
            assert check_u8(uw.enhance(u8, strategy=strategy), orc.enhance_u8(u8, strategy), f"{H}x{W} strategy {strategy}") == 0


def test_enhance_640x480_canonical(uw, orc):
    rng = np.random.default_rng(1000)
    yy, xx = np.mgrid[0:480, 0:640]
    field = 0.5 + 0.25 * (np.sin(xx / 61.0) * np.cos(yy / 47.0) + 0.5 * np.sin((xx + 2 * yy) / 113.0)) / 1.5
    f = field[:, :, None] * np.array([0.45, 0.85, 0.80]) + rng.normal(0, 0.02, (480, 640, 3))
    u8 = np.clip(np.floor(255 * f), 0, 255).astype(np.uint8)
    got = uw.enhance(u8)
    assert check_u8(got, orc.enhance_u8(u8, 2), "canonical 640x480") == 0


def test_batch_equals_singles_and_torch_path(uw):
    import torch

    rng = np.random.default_rng(77)
    batch = rng.integers(0, 256, (3, 61, 83, 3), dtype=np.uint8)
    batch[1, :, :, 1] = np.minimum(batch[1, :, :, 1].astype(int) + 70, 255)
    out = uw.enhance(batch)
    for b in range(3):
        assert np.array_equal(out[b], uw.enhance(batch[b]))
    t = torch.from_numpy(batch).cuda()
    out_t = uw.enhance(t)
    assert out_t.is_cuda and np.array_equal(out_t.cpu().numpy(), out)


def test_cast_correct_normal_is_identity(uw):
    rng = np.random.default_rng(78)
    u8 = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)  # neutral noise: detect_image_type -> "normal"
    assert np.array_equal(uw.enhance(u8, cast_correct=True), uw.enhance(u8, cast_correct=False))


def test_float_surface_mirrors_reference_api(uw, orc):
    rng = np.random.default_rng(79)
    u8 = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    u8[:, :, 1] = np.minimum(u8[:, :, 1].astype(int) + 80, 255)
    x = orc.normalise_u8(u8)
    kind = uw.detect_image_type(x)
    assert kind == orc.classify_cast(x) == "greenish"
    xc = uw.color_correction(x, kind)
    assert np.array_equal(xc, orc.correct_cast(x, kind))
    assert uw.color_correction(x, "normal") is x
    y = uw.SixStrategies.strategy2_medium_dehazing(xc)
    want = orc.SixStrategyOracle.strategy(2, orc.correct_cast(x, kind))
    assert y.dtype == np.float32 and np.array_equal(y, want)
    r = rng.random((40, 44, 3)).astype(np.float32)  # not u8-derived: the general float path (see the test further down)
    assert np.abs(uw.SixStrategies.strategy2_medium_dehazing(r) - orc.SixStrategyOracle.strategy(2, r)).max() <= 2e-7
    with pytest.raises(ValueError):
        uw.EnhancementStrategies.apply_strategy(x, "no_such_strategy", {})


# ------------------------------------------------------------------ enhancement_strategies.py (dict) surface
@pytest.mark.parametrize("name", ["strong_dehazing", "medium_dehazing", "light_enhancement", "clahe_enhancement",
                                  "histogram_equalization"])
def test_dict_surface_matches_oracle(uw, orc, name):
    """apply_strategy(img, name, Config.STRATEGIES[name]) (ES:477-508, config.py:28-75) and the in-code defaults."""
    frames = frames_for_tests(np.random.default_rng(4242))
    frames.pop("tiny_5x7")
    ES = orc.DictStrategyOracle
    for params in (orc.CONFIG_STRATEGIES[name], {}):
        for tag, u8 in frames.items():
            x = orc.normalise_u8(u8)
            want = ES.run(x, name, params)  # float64 image of the reference
            got = uw.EnhancementStrategies.apply_strategy(x, name, params)
            assert got.dtype == np.float64 and got.shape == want.shape  # ES:247,307,345: the reference returns float64
            # u8 image as main.py:155 makes it; the device also quantises in float64
            dev = uw.get_device(0)
            over = {}
            for key, field in (("omega", "omega"), ("guided_radius", "gf_ksize"), ("L_low", "L_low"), ("L_high", "L_high"),
                               ("clip_limit", "clip_limit"), ("gamma", "gamma")):
                if key in params:
                    over[field] = params[key]
            over["apply_gamma"] = int(bool(params.get("apply_gamma", False)))
            p = dev.params(1, {"strong_dehazing": 0, "medium_dehazing": 1, "light_enhancement": 2, "clahe_enhancement": 3,
                               "histogram_equalization": 4}[name], **over)
            out_u8, out_f = dev.enhance_u8(dev.tensor(u8[None]), p, want_float=True)
            d = np.abs(out_u8[0].cpu().numpy().astype(int) - (want * 255).astype(np.uint8).astype(int))
            assert d.max() <= 1, f"{name} on {tag}: {d.max()} LSB"
            dehaze = name in ("strong_dehazing", "medium_dehazing", "light_enhancement")
            if not params.get("apply_gamma", False):  # without pow everything is reproduced bit for bit ...
                assert d.max() == 0, f"{name} on {tag}: {np.count_nonzero(d)} bytes differ"
                assert np.array_equal(out_f[0].cpu().numpy(), want.astype(np.float32))  # the float32 copy of uwie_enhance_u8
                if dehaze:  # ... in float64 with the exact-order guided filter; the fused one is within its 1e-11 on t
                    p.gf_exact = 1
                    assert np.array_equal(dev.enhance_u8_f64(dev.tensor(u8[None]), p)[1][0].cpu().numpy(), want)
                    assert np.abs(got - want).max() < 1e-9
                else:
                    assert np.array_equal(got, want)
                # what a caller following main.py:155 computes from the returned image is the reference's byte
                assert np.array_equal((got * 255).astype(np.uint8), (want * 255).astype(np.uint8))
            else:
                assert np.abs(got - want).max() < 1e-9  # float64 pow: a few ulp between libraries
                dq = np.abs((got * 255).astype(np.uint8).astype(int) - (want * 255).astype(np.uint8).astype(int))
                assert dq.max() <= 1


def test_dict_surface_select_and_store_modes(uw, orc):
    """The dict surface's dehazing strategies take their float64 percentiles from the linear-digit selection on an image
    that is recomputed per sweep, with the target bins predicted from a sample (default); prediction off or missing,
    the stored-plane mode, the forced fallback to the generic key sweeps (tiny
    candidate lists: flagged planes are written out first) and the generic sweeps alone must give the same floats."""
    rng = np.random.default_rng(515)
    noisy = rng.integers(0, 256, (150, 210, 3), dtype=np.uint8)
    flatish = np.empty((330, 310, 3), np.uint8)
    flatish[:] = (90, 140, 180)
    flatish[::7, ::5] = rng.integers(0, 256, flatish[::7, ::5].shape, dtype=np.uint8)
    odd = rng.integers(0, 256, (131, 203, 3), dtype=np.uint8)
    ES = orc.DictStrategyOracle
    for name in ("strong_dehazing", "medium_dehazing", "light_enhancement"):
        for tag, u8 in (("noisy", noisy), ("flatish", flatish), ("odd", odd)):
            x = orc.normalise_u8(u8)
            want = ES.run(x, name, {})
            for env in ({}, {"restore_store": 1}, {"lin_cap": 16}, {"lin_cap": 16, "restore_store": 1},
                        {"select_generic": 1}, {"lin_no_predict": 1}, {"lin_predict_shift": 400},
                        {"lin_predict_shift": 400, "restore_store": 1}):
                with uw.get_device().tuning(**env):
                    got = uw.EnhancementStrategies.apply_strategy(x, name, {})
                # float64 image: the fused guided filter's tolerance on t (1e-11) shows; identical after the float32 rounding
                assert np.abs(got - want).max() < 1e-9 and np.array_equal(got.astype(np.float32), want.astype(np.float32)), (
                    name, tag, env, int((got.astype(np.float32) != want.astype(np.float32)).sum()))


def test_dict_surface_error_behaviour(uw, orc):
    rng = np.random.default_rng(5)
    x = orc.normalise_u8(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8))
    with pytest.raises(ValueError):
        uw.EnhancementStrategies.apply_strategy(x, "weak_dehazing", {})  # commented out in the reference (ES:494-496)
    # ES:503-508 swallows failures INSIDE a strategy and returns the input (here: a window wider than the frame)
    assert uw.EnhancementStrategies.apply_strategy(x, "strong_dehazing", {"guided_radius": 4000}) is x
    # ... but an image this build has no device path for is refused, never passed through as a success
    with pytest.raises(uw.UnsupportedInputError):
        uw.EnhancementStrategies.apply_strategy((x * 255).astype(np.int32), "strong_dehazing", {})
    with pytest.raises(uw.UnsupportedInputError):
        uw.SixStrategies.strategy2_medium_dehazing(rng.random((16, 16, 3)))  # float64 on the float32 surface


@pytest.mark.parametrize("strategy", [1, 2, 3])
def test_exact_order_guided_filter_mode(uw, orc, strategy):
    """gf_exact=1 (cv2.boxFilter's running-sum order) and the default fused filter give the same u8 image here."""
    frames = frames_for_tests(np.random.default_rng(4242))
    for name in ("green_120x160", "hazy_97x131", "noise_61x83"):
        u8 = frames[name]
        want = orc.enhance_u8(u8, strategy)
        assert np.array_equal(uw.enhance(u8, strategy=strategy, gf_exact=1), want)
        assert check_u8(uw.enhance(u8, strategy=strategy, gf_exact=0), want, name) == 0


# ------------------------------------------------------------------ BASELINE.json sizes
def _underwater(rng, H, W, gains):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    field = 0.55 + 0.25 * (np.sin(xx / (W / 9.0)) * np.cos(yy / (H / 7.0)) + 0.5 * np.sin((xx + 2 * yy) / (W / 5.0))) / 1.5
    f = field[:, :, None] * np.array(gains, np.float32)[None, None, :] + rng.normal(0, 0.02, (H, W, 3)).astype(np.float32)
    return np.clip(np.floor(255 * f), 0, 255).astype(np.uint8)


def test_config1_1080p_single_frame_matches_oracle(uw, orc):
    """BASELINE.json configs[1]: 1920x1080 RGB, batch 1, full WB + DCP + guided filter + CLAHE."""
    u8 = _underwater(np.random.default_rng(1001), 1080, 1920, (0.45, 0.85, 0.80))
    got = uw.enhance(u8)
    want = orc.enhance_u8(u8, 2)
    n = check_u8(got, want, "1080p")
    assert n <= 8, f"{n} bytes differ by 1 LSB"  # default (fused) guided filter: <= 1 LSB, practically identical
    assert np.array_equal(uw.enhance(u8, gf_exact=1), want)  # exact-order mode: bit for bit


def test_parameter_checks_and_merged_fan_out_workspace(uw, orc):
    """np.percentile raises for q outside [0, 100] (and NaN): the C ABI refuses such parameter sets instead of casting a
    negative rank to unsigned.  uwie_enhance_all_u8 with caller-supplied sets sizes its workspace for the most demanding
    of them (here: strategy 2 in exact-order mode and strategy 4 with a wider tile grid)."""
    import ctypes

    from underwater_image_enhancement_amd import _lib

    u8 = _underwater(np.random.default_rng(77), 96, 128, (0.45, 0.85, 0.80))
    for bad in ({"L_low": -1.0}, {"L_high": 100.5}, {"L_low": float("nan")}, {"inter_dtype": 7}):
        with pytest.raises(_lib.UwieError):
            uw.enhance(u8, strategy=2, **bad)
    with pytest.raises(_lib.UwieError):
        uw.enhance(u8, strategy=3, wb_percentile=101.0)
    dev = uw.get_device()
    import torch

    p6 = (_lib.UwieParams * 6)()
    for k in range(6):
        p6[k] = dev.params(_lib.SURFACE_SIX, k + 1)
    p6[1].gf_exact = 1
    p6[3].tiles_x = p6[3].tiles_y = 12
    frames = dev.tensor(u8[None])
    need = dev.lib.uwie_workspace_bytes_all(1, 96, 128, ctypes.cast(p6, ctypes.c_void_p))
    assert need >= dev.lib.uwie_workspace_bytes(1, 96, 128, ctypes.byref(p6[1]))
    ws = dev.workspace(need)
    out = dev.empty((6, 1, 96, 128, 3), torch.uint8)
    _lib.check(dev.lib.uwie_enhance_all_u8(dev._ctx, ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, 1, 96,
                                           128, ctypes.cast(p6, ctypes.c_void_p), ctypes.c_void_p(ws.data_ptr()), need, dev.stream()))
    got = out.cpu().numpy()
    assert np.array_equal(got[1, 0], orc.enhance_u8(u8, 2))
    assert np.array_equal(got[3, 0], uw.enhance(u8, strategy=4, tiles_x=12, tiles_y=12))
    # too small a workspace is an error, not an out-of-bounds write
    rc = dev.lib.uwie_enhance_all_u8(dev._ctx, ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, 1, 96, 128,
                                     ctypes.cast(p6, ctypes.c_void_p), ctypes.c_void_p(ws.data_ptr()),
                                     dev.lib.uwie_workspace_bytes(1, 96, 128, ctypes.byref(p6[0])) // 2, dev.stream())
    assert rc != 0


def test_config2_4k_batch64(uw, orc):
    """BASELINE.json configs[2] at its full size: 64 frames of 3840x2160 in one call (the split-ring guided kernel, the
    wide quadtree levels and every other stage at the launch shapes bench.py times).  Size-independent properties: the
    batch equals the single-frame calls on sampled frames (first, last, a greenish, a bluish and a noise frame), one frame
    equals the oracle, and the workspace the library asks for is enough (the call would fail otherwise)."""
    import torch

    dev = uw.get_device()
    B, H, W = 64, 2160, 3840
    g = torch.Generator(device=dev.torch_device).manual_seed(642)
    yy = torch.arange(H, device=dev.torch_device, dtype=torch.float32)[:, None]
    xx = torch.arange(W, device=dev.torch_device, dtype=torch.float32)[None, :]
    frames = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev.torch_device)
    for b in range(B):
        ph = torch.rand(4, generator=g, device=dev.torch_device) * 6.283
        field = 0.55 + 0.125 * (torch.sin(xx / (W / 9.0) + ph[0]) * torch.cos(yy / (H / 7.0) + ph[1])
                                + 0.5 * torch.sin((xx + 2 * yy) / (W / 5.0) + ph[2]))
        gains = (0.45, 0.85, 0.80) if b % 2 == 0 else (0.45, 0.75, 0.90)
        for c in range(3):
            ch = field * gains[c] + torch.randn((H, W), generator=g, device=dev.torch_device) * 0.02
            frames[b, :, :, c] = torch.clamp(torch.floor(ch * 255.0), 0, 255).to(torch.uint8)
    frames[37] = torch.randint(0, 256, (H, W, 3), generator=g, device=dev.torch_device, dtype=torch.uint8)
    out = uw.enhance(frames)
    assert out.shape == frames.shape and out.dtype == torch.uint8
    for b in (0, 1, 37, 62, 63):
        assert torch.equal(out[b], uw.enhance(frames[b:b + 1])[0]), f"frame {b} of the batch differs from its single call"
    u8 = frames[2].cpu().numpy()
    assert check_u8(out[2].cpu().numpy(), orc.enhance_u8(u8, 2), "frame 2 of the 4K x 64 batch") == 0
    del frames, out
    torch.cuda.empty_cache()


def test_config2_4k_frame_matches_oracle_and_batch_is_invariant(uw, orc):
    """BASELINE.json configs[2] frame size (3840x2160): one frame against the oracle, then the size-independent
    properties on a batch: batch == singles, frame order does not matter, a 'normal' cast is the identity."""
    rng = np.random.default_rng(1002)
    frames = np.stack([_underwater(rng, 2160, 3840, (0.45, 0.85, 0.80)), _underwater(rng, 2160, 3840, (0.45, 0.75, 0.90)),
                       rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)])
    out = uw.enhance(frames)
    assert check_u8(out[0], orc.enhance_u8(frames[0], 2), "4K frame") <= 32
    for b in range(3):
        assert np.array_equal(out[b], uw.enhance(frames[b]))
    assert np.array_equal(uw.enhance(frames[::-1].copy())[::-1], out)
    assert np.array_equal(uw.enhance(frames[2], cast_correct=False), out[2])  # neutral noise is classified "normal"


def test_enhance_all_equals_the_six_single_strategy_calls(uw, orc):
    """uwie_enhance_all_u8 (N1: the batch driver's fan-out, six_stadigy.py:398-431) shares cast detection, gray plane
    and quadtree across strategies; every plane must equal the single-strategy call bit for bit, the image types must
    be detect_image_type's, and one frame is checked against the oracle directly."""
    frames = frames_for_tests(np.random.default_rng(515))
    frames.pop("tiny_5x7")
    for name, u8 in frames.items():
        outs, types = uw.enhance_all(u8)
        assert list(outs) == [n for n, _ in uw.DRIVER_STRATEGIES]
        assert types == [orc.classify_cast(orc.normalise_u8(u8))]
        for k, sname in enumerate(outs, start=1):
            assert np.array_equal(outs[sname], uw.enhance(u8, strategy=k)), (name, sname)
    name, u8 = next(iter(frames.items()))
    outs, _ = uw.enhance_all(u8)
    for k, sname in enumerate(outs, start=1):
        check_u8(outs[sname], orc.enhance_u8(u8, k), f"enhance_all {sname} on {name}")
    rng = np.random.default_rng(9)
    batch = rng.integers(0, 256, (3, 70, 90, 3), dtype=np.uint8)
    batch[1, :, :, 0] //= 3  # a frame with a cast
    outs, types = uw.enhance_all(batch)
    for b in range(3):
        one, t1 = uw.enhance_all(batch[b])
        assert t1 == [types[b]]
        for sname in outs:
            assert np.array_equal(outs[sname][b], one[sname])
    _, rows, stats = uw.process_batch(batch, ["a.png", "b.png", "c.png"])
    assert len(rows) == 18 and stats["successful_outputs"] == 18 and sum(stats["image_types"].values()) == 3


def test_stream_enhancer_matches_direct_calls(uw):
    """Host-memory streaming (configs[4]): chunks through pinned buffers on three streams equal direct enhance() calls,
    including a short last chunk and more chunks than ring slots."""
    rng = np.random.default_rng(88)
    frames = rng.integers(0, 256, (11, 60, 84, 3), dtype=np.uint8)
    frames[3, :, :, 0] //= 4
    want = uw.enhance(frames)
    se = uw.StreamEnhancer(60, 84, chunk=3, depth=3)
    outs = [o.numpy().copy() for o in se.run(frames[i:i + 3] for i in range(0, 11, 3))]
    assert [o.shape[0] for o in outs] == [3, 3, 3, 2]
    assert np.array_equal(np.concatenate(outs), want)
    # zero-copy producer interface, ring reuse
    got = []
    for i in range(5):
        if i >= 2:
            got.append(se.result().numpy().copy())
        se.input_slot(i)[:2].copy_(torch_from(frames[2 * i:2 * i + 2]))
        se.submit_slot(i, 2)
    while se._pending:
        got.append(se.result().numpy().copy())
    assert np.array_equal(np.concatenate(got), want[:10])


def torch_from(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a))


def test_sub_batch_streams_and_profile_filter(uw):
    """Tuning streams = 2 / 3 (sub-batches on internal streams, joined on the caller's stream) must not change a byte, for
    every surface; the profiler's name filter records only the named kernel."""
    from underwater_image_enhancement_amd import _lib

    rng = np.random.default_rng(123)
    frames = rng.integers(0, 256, (7, 66, 98, 3), dtype=np.uint8)
    frames[2, :, :, 0] //= 3
    want = {k: uw.enhance(frames, strategy=k) for k in (1, 3, 5)}
    dev = uw.get_device()
    for n in (2, 3):
        with dev.tuning(streams=n):
            for k, w in want.items():
                assert np.array_equal(uw.enhance(frames, strategy=k), w), (n, k)
    dev.profile(True)
    uw.enhance(frames, strategy=2)
    every = dev.profile_rows()
    assert "k_guided_pipe" in every and len(every) > 10
    dev.profile(True, only="k_guided_pipe")
    uw.enhance(frames, strategy=2)
    rows = dev.profile_rows()
    dev.profile(False)
    assert list(rows) == ["k_guided_pipe"] and rows["k_guided_pipe"][1] == 1
    assert _lib.load().uwie_profile_filter(dev._ctx, None) == 0


def test_select_paths_agree(uw, orc):
    """The percentile selection of strategies 1-3 has three routes: linear first digit with collected candidates (default),
    its fallback to the generic sweeps when a candidate list overflows (forced here with a tiny list capacity, and hit for
    real by a nearly constant frame), and the generic three-digit sweeps alone; strategies 1-2 run them either on the
    the restored image recomputed per sweep (default) or on stored planes (tuning restore_store).  All must give the
    oracle's bytes."""
    rng = np.random.default_rng(404)
    noisy = rng.integers(0, 256, (150, 210, 3), dtype=np.uint8)
    flatish = np.empty((330, 310, 3), np.uint8)  # > 65536 pixels, almost all of one colour: one heavy interior bin
    flatish[:] = (90, 140, 180)
    flatish[::7, ::5] = rng.integers(0, 256, flatish[::7, ::5].shape, dtype=np.uint8)
    odd = rng.integers(0, 256, (131, 203, 3), dtype=np.uint8)  # pixel count and tile widths not multiples of 4
    yy, xx = np.mgrid[0:520, 0:1000]  # > 262144 pixels: the prediction works from a subsample of the rows
    wide = np.stack([40 + 0.1 * xx + 0.2 * yy, 60 + 0.15 * xx, 200 - 0.1 * yy], -1) + rng.normal(0, 12, (520, 1000, 3))
    wide = np.clip(wide, 0, 255).astype(np.uint8)
    big = np.empty((700, 720, 3), np.uint8)  # a block meets > 512 candidates of one bin in one step: LDS stage overflow
    big[:] = (60, 120, 200)
    big[::11, ::13] = rng.integers(0, 256, big[::11, ::13].shape, dtype=np.uint8)
    want_big = orc.enhance_u8(big, 2)
    dev = uw.get_device()
    # (store, rank): stored planes or recomputation; for the latter the histogram sweep (the default below 16 MP) and round 4's
    # rank-counting sweep (rank_sweep = 2 forces it onto these small frames), each through every fallback below
    for store, rank in ((0, 1), (0, 2), (1, 1)):
        with dev.tuning(restore_store=store, rank_sweep=rank):
            for name, u8 in (("noisy", noisy), ("flatish", flatish), ("odd", odd), ("wide", wide)):
                for k in (1, 2, 3):
                    want = orc.enhance_u8(u8, k)
                    check_u8(uw.enhance(u8, strategy=k), want, f"default select, strategy {k} on {name}, store={store}")
                    with dev.tuning(lin_cap=16):
                        check_u8(uw.enhance(u8, strategy=k), want, f"forced fallback, strategy {k} on {name}, store={store}")
                    with dev.tuning(select_generic=1):
                        check_u8(uw.enhance(u8, strategy=k), want, f"generic sweeps only, strategy {k} on {name}, store={store}")
                    # the producer files the predicted windows (two, or strategy 3's four): off, still covering, missing
                    for knob, val in (("lin_no_predict", 1), ("lin_predict_shift", 2), ("lin_predict_shift", 400), ("lin_predict3", 1)):
                        with dev.tuning(**{knob: val}):
                            check_u8(uw.enhance(u8, strategy=k), want, f"{knob}={val}, strategy {k} on {name}, store={store}")
            check_u8(uw.enhance(big, strategy=2), want_big, f"stage overflow, strategy 2 on big flat frame, store={store}")
    batch = np.stack([noisy[:120, :200], flatish[:120, :200], noisy[30:150, 10:210]])
    assert np.array_equal(uw.enhance(batch, strategy=2), np.stack([uw.enhance(f, strategy=2) for f in batch]))


# ------------------------------------------------------------------ general (not u8-derived) float images
def test_general_float_images_on_both_surfaces(uw, orc):
    """The reference's functions take any float image in [0, 1]; its own harnesses feed np.random.rand
    (enhancement_strategies.py:516,526; example_usage.py:27-31,44-48,112).  Not u8-derived images take the general float
    path (uwie_enhance_f32 / _f64): float64 and float32 through the five dict strategies, float32 through the six
    six_stadigy strategies, against the oracle.  Without pow and with the exact-order guided filter every float is
    reproduced bit for bit; the fused guided filter (default on float32 images) is within its 1e-11 on t, and pow within a
    few ulp between libraries."""
    rng = np.random.default_rng(20240516)
    x64 = rng.random((256, 256, 3))
    x32 = rng.random((96, 131, 3)).astype(np.float32)
    hazy = (rng.random((120, 88, 3)) * 0.7 + 0.15).astype(np.float32)  # example_usage.py:112
    ES = orc.DictStrategyOracle
    for name in ("strong_dehazing", "medium_dehazing", "light_enhancement", "clahe_enhancement", "histogram_equalization"):
        for params in (orc.CONFIG_STRATEGIES[name], {}):
            for x in (x64, x32, hazy):
                want = ES.run(x, name, params)
                got = uw.EnhancementStrategies.apply_strategy(x, name, params)
                assert got.dtype == np.float64 and got.shape == want.shape, (name, x.dtype)
                tol = 1e-9 if (params.get("apply_gamma", False) or (x.dtype == np.float32 and "dehaz" in name or name == "light_enhancement")) else 0.0
                err = np.abs(got - want).max()
                assert err <= tol, (name, params, x.dtype, err)
                dq = np.abs((got * 255).astype(np.uint8).astype(int) - (want * 255).astype(np.uint8).astype(int))
                assert dq.max() <= (1 if tol else 0), (name, x.dtype, int(dq.max()))
    S6 = orc.SixStrategyOracle
    fns = (uw.SixStrategies.strategy1_strong_dehazing, uw.SixStrategies.strategy2_medium_dehazing,
           uw.SixStrategies.strategy3_light_dehazing, uw.SixStrategies.strategy4_clahe_enhancement,
           uw.SixStrategies.strategy5_white_balance, uw.SixStrategies.strategy6_histogram_eq)
    for k, fn in enumerate(fns, start=1):
        for x in (x32, hazy):
            want = S6.strategy(k, x)
            got = fn(x)
            assert got.dtype == np.float32 and got.shape == want.shape
            assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2e-7, (k, np.abs(got - want).max())
            dq = np.abs((got * 255).astype(np.uint8).astype(int) - (want * 255).astype(np.uint8).astype(int))
            assert dq.max() <= 1, (k, int(dq.max()))
    # cast detection / correction on a general float image
    g = x32.copy()
    g[:, :, 1] = np.clip(g[:, :, 1] + 0.2, 0, 1)
    assert uw.detect_image_type(g) == orc.classify_cast(g) == "greenish"
    assert uw.detect_image_type(x32) == orc.classify_cast(x32)
    assert np.array_equal(uw.color_correction(g, "greenish"), orc.correct_cast(g, "greenish"))
    with pytest.raises(uw.UnsupportedInputError):
        uw.SixStrategies.strategy2_medium_dehazing(x64)  # six_stadigy.py works on float32 frames


def test_select_best_equals_the_labelling_loop_of_main_py(uw, orc):
    """main.py:118-146 -- the five Config.STRATEGIES through apply_strategy, comprehensive_assessment with
    Config.QUALITY_WEIGHTS on each result, the first maximum wins -- as ONE device call (uwie_select_best_u8; the three
    dehazing strategies share one quadtree) against the same loop through the oracle: every strategy's bytes (<= 1 LSB where
    gamma's pow is involved, identical otherwise), every total within the quality scores' 2e-3, and the same winner wherever
    the oracle's two best totals are further apart than that."""
    from test_gpu_configs import underwater

    rng = np.random.default_rng(118)
    frames = np.stack([underwater(rng, 240, 320, (0.45, 0.85, 0.80)), underwater(rng, 240, 320, (0.45, 0.75, 0.90)),
                       rng.integers(0, 256, (240, 320, 3), dtype=np.uint8),
                       np.floor(255 * (rng.random((240, 320, 3)) * 0.7 + 0.15)).astype(np.uint8)])
    names, images, scores, every = uw.select_best(frames, return_all=True)
    keys = list(uw.CONFIG_STRATEGIES)
    assert list(every) == [uw.CONFIG_STRATEGIES[k]["name"] for k in keys] and images.shape == frames.shape
    for b, u8 in enumerate(frames):
        x = orc.normalise_u8(u8)  # main.py:108
        want_scores, want_imgs = {}, {}
        for k in keys:
            params = {kk: v for kk, v in uw.CONFIG_STRATEGIES[k].items() if kk != "name"}
            enhanced = orc.DictStrategyOracle.run(x, k, params)
            total, _ = orc.quality_assessment(enhanced, weights=uw.CONFIG_QUALITY_WEIGHTS)
            name = uw.CONFIG_STRATEGIES[k]["name"]
            want_scores[name] = float(total)
            want_imgs[name] = (enhanced * 255).astype(np.uint8)  # main.py:155
            d = np.abs(every[name][b].astype(int) - want_imgs[name].astype(int))
            assert d.max() <= (1 if params.get("apply_gamma") else 0), (b, name, int(d.max()))
            assert abs(scores[b][name] - want_scores[name]) <= 2e-3 + 0.05 * int(d.max() > 0), (b, name, scores[b][name], want_scores[name])
        ranked = sorted(want_scores.values(), reverse=True)
        want_best = max(want_scores, key=want_scores.get)  # main.py:145
        if ranked[0] - ranked[1] > 0.2:
            assert names[b] == want_best, (b, names[b], want_best, want_scores)
        assert np.array_equal(images[b], every[names[b]][b])
    # a single frame, a subset of the strategies, custom weights: same protocol
    sub = {k: uw.CONFIG_STRATEGIES[k] for k in ("clahe_enhancement", "light_enhancement")}
    name1, img1, sc1 = uw.select_best(frames[0], strategies=sub, weights={"contrast": 1.0})
    assert name1 in ("CLAHEEnhancement", "LightEnhancement") and img1.shape == frames[0].shape and set(sc1) == {"CLAHEEnhancement", "LightEnhancement"}
    assert np.array_equal(img1, every[name1][0]) and name1 == max(sc1, key=sc1.get)


def test_fused_transmission_route_gives_the_same_bytes(uw, orc):
    """Tuning gf_fuse = 1 (k_guided_split8, round 4): the first half of estimate_transmission (S6:170-174 / ES:221-225) is
    evaluated inside the guided filter from the frame's bytes -- the same float32 operations as k_trans_init, so the bytes are
    those of the default route (and the oracle's) on both surfaces, for the float64 and the float32 transmission.  gf_bands
    forces the split kernels onto frames this small; an edge strip on each side and interior strips are all there."""
    from underwater_image_enhancement_amd import _lib
    from test_gpu_configs import underwater

    rng = np.random.default_rng(170)
    dev = uw.get_device()
    frames = [underwater(rng, 300, 520, (0.45, 0.85, 0.80)), rng.integers(0, 256, (260, 384, 3), dtype=np.uint8)]
    for u8 in frames:
        x = orc.normalise_u8(u8)
        with dev.tuning(gf_bands=2):
            base2 = uw.enhance(u8, strategy=2)
            base2f = uw.enhance(u8, strategy=2, inter_dtype=_lib.INTER_F32T)
            based = uw.EnhancementStrategies.apply_strategy(x, "strong_dehazing", orc.CONFIG_STRATEGIES["strong_dehazing"])
            with dev.tuning(gf_fuse=1):
                assert np.array_equal(uw.enhance(u8, strategy=2), base2)
                assert np.array_equal(uw.enhance(u8, strategy=2, inter_dtype=_lib.INTER_F32T), base2f)
                assert np.array_equal(uw.EnhancementStrategies.apply_strategy(x, "strong_dehazing", orc.CONFIG_STRATEGIES["strong_dehazing"]), based)
        assert np.array_equal(base2, orc.enhance_u8(u8, 2))
