"""Round 4, tuning entry_fuse (frames with W % 8 == 0): the level-0 quadrant histograms of the quadtree are counted by cast
detection's chunk pass (k_chunk_hist_quad + k_quad_hist_reduce, six_stadigy.py:292-302 and :116-156 read the frame once) and
the gray plane (six_stadigy.py:149,177) is written by the level-0 Canny pre-pass (k_gray_strong).  Both routes must give the
bytes of the separate passes of rounds 3 - 4 (entry_fuse = 0) and of the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uw():
    import underwater_image_enhancement_amd as uw

    return uw


def _frames(rng, B, H, W, style):
    yy, xx = np.mgrid[0:H, 0:W]
    out = []
    for i in range(B):
        if style == "water":  # smooth water with sensor noise: no strong pixel in most quadrants
            f = 0.5 + 0.25 * np.sin(xx / (17.0 + i)) * np.cos(yy / (11.0 + 2 * i))
            img = f[:, :, None] * np.array([0.45, 0.85, 0.80]) + rng.normal(0, 0.02, (H, W, 3))
        elif style == "blue":
            f = 0.4 + 0.2 * np.cos(xx / 23.0 + i) * np.sin(yy / 7.0)
            img = f[:, :, None] * np.array([0.35, 0.6, 1.0]) + rng.normal(0, 0.03, (H, W, 3))
        elif style == "noise":  # strong pixels everywhere
            img = rng.random((H, W, 3))
        else:  # "edges": flat quadrants with a few hard steps, one of them on the last column / row of a quadrant
            img = np.full((H, W, 3), 0.3) + rng.normal(0, 0.004, (H, W, 3))
            img[:, W // 2 - 1] += 0.6 * (i % 2)
            img[H // 2 :, : W // 4] += 0.5
            img[H // 2 - 1, W // 2 :] += 0.7 * ((i + 1) % 2)
            img[0, 0] = 1.0
        out.append(np.clip(np.floor(255 * img), 0, 255).astype(np.uint8))
    return np.stack(out)


SHAPES = [(200, 264), (258, 512), (333, 1000), (540, 960)]


@pytest.mark.parametrize("style", ["water", "blue", "noise", "edges"])
def test_gray_plane_and_quadtree_of_the_fused_prepass(uw, style):
    """k_gray_strong against k_quant_gray (the gray plane) and against the separate passes (atmospheric light and the
    quadtree's scores), every cast kind, quadrants whose borders matter (steps on their last column / row)."""
    dev = uw.get_device()
    rng = np.random.default_rng({"water": 1, "blue": 2, "noise": 3, "edges": 4}[style])
    for H, W in SHAPES:
        u8 = _frames(rng, 3, H, W, style)
        frames = dev.tensor(u8)
        for kinds in (None, [0, 1, 2], [2, 2, 1]):
            kk = None if kinds is None else dev.tensor(np.array(kinds, np.int32))
            _, want_gray = dev.transmission_init(frames, dev.tensor(np.full((3, 3), 0.8, np.float32)), kk)
            got = {}
            for fuse in (0, 1):
                with dev.tuning(entry_fuse=fuse):
                    A, tr, gray = dev.atmospheric_light(frames, kk, trace=True, want_gray=True)
                    A2 = dev.atmospheric_light(frames, kk)  # without the trace: levels decided from the histograms
                    dev.check_status()
                got[fuse] = (A.cpu().numpy(), tr, gray.cpu().numpy(), A2.cpu().numpy())
            assert np.array_equal(got[1][2], want_gray.cpu().numpy()), (style, H, W, kinds)
            assert np.array_equal(got[0][2], want_gray.cpu().numpy())
            assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[1][3], got[0][3]), (style, H, W, kinds)
            assert got[1][1].tobytes() == got[0][1].tobytes(), (style, H, W, kinds)
            assert np.array_equal(got[1][0], got[1][3])


@pytest.mark.parametrize("strategy", [1, 2, 3])
def test_enhance_with_the_quadrant_histograms_of_the_chunk_pass(uw, strategy):
    """Whole pipeline: cast detection + quadtree through k_chunk_hist_quad / k_quad_hist_reduce / k_gray_strong against the
    separate passes, and against the oracle on one frame per shape.  Shapes: the middle row inside a chunk, on a chunk
    boundary (256 x 512: 128 rows * 512 = 4 chunks), chunks shorter than a row's half."""
    from oracle import uwie_oracle as orc

    from underwater_image_enhancement_amd import _lib

    dev = uw.get_device()
    rng = np.random.default_rng(77 + strategy)
    for (H, W), style in zip([(200, 264), (256, 512), (301, 1000), (540, 960), (130, 4096)], ["water", "blue", "noise", "edges", "water"]):
        u8 = _frames(rng, 2, H, W, style)
        frames = dev.tensor(u8)
        p = dev.params(_lib.SURFACE_SIX, strategy)
        got = {}
        for fuse in (0, 1, 2):  # 1: gray plane by the chunk pass (guessed kind), 2: by the quadtree's pre-pass
            with dev.tuning(entry_fuse=fuse):
                out, _ = dev.enhance_u8(frames, p)
                dev.check_status()
            got[fuse] = out.cpu().numpy()
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2]), (H, W, style)
        if H * W <= 300000:
            want = orc.enhance_u8(u8[0], strategy)
            assert np.abs(got[1][0].astype(int) - want.astype(int)).max() <= 1, (H, W, style)


def test_near_tie_quadrants_take_the_exact_kernels_after_the_fused_pass(uw):
    """Quadrants that only differ in a few pixels: the interval test of k_q_decide cannot decide level 0, so the exact kernels
    run behind the fused histograms (the q_hist = 3 checking route verifies every interval against the exact score)."""
    from underwater_image_enhancement_amd import _lib

    dev = uw.get_device()
    rng = np.random.default_rng(5)
    H, W = 384, 512
    tile = np.floor(255 * (0.4 + 0.2 * rng.random((H // 2, W // 2, 3)))).astype(np.uint8)
    u8 = np.tile(tile, (2, 2, 1))[None].copy()
    u8[0, 3, 5, 1] += 1
    frames = dev.tensor(u8)
    p = dev.params(_lib.SURFACE_SIX, 2)
    outs = []
    for fuse, qh in ((0, 1), (1, 1), (1, 3), (1, 2)):
        with dev.tuning(entry_fuse=fuse, q_hist=qh):
            out, _ = dev.enhance_u8(frames, p)
            dev.check_status()
        outs.append(out.cpu().numpy())
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_a_wrong_guess_of_the_cast_kind_is_repaired(uw):
    """The chunk pass writes the gray plane for a cast kind guessed from 2048 strided pixels before the real decision exists.
    Frames built so that the guess is wrong both ways (the sampled pixels green on a neutral frame; neutral on a green one):
    k_quant_gray writes their planes again, the output equals the separate passes and the oracle."""
    from oracle import uwie_oracle as orc

    from underwater_image_enhancement_amd import _lib

    dev = uw.get_device()
    rng = np.random.default_rng(9)
    H, W = 264, 512
    npx = H * W
    stride = npx // 2048
    idx = np.arange(2048) * stride
    neutral = np.floor(255 * (0.45 + 0.1 * rng.random((H, W, 3)))).astype(np.uint8)
    green = neutral.copy()
    green[:, :, 1] = np.minimum(255, green[:, :, 1].astype(int) + 60)
    a = neutral.copy().reshape(-1, 3)
    a[idx] = (100, 170, 100)  # guess: greenish; decision: normal (1.5 % of the pixels move the mean by 0.004)
    b = green.copy().reshape(-1, 3)
    b[idx] = (128, 128, 128)  # guess: normal; decision: greenish
    u8 = np.stack([a.reshape(H, W, 3), b.reshape(H, W, 3), neutral, green])
    frames = dev.tensor(u8)
    kind, _ = dev.cast_classify(frames)
    assert kind.cpu().numpy().tolist() == [0, 1, 0, 1]
    p = dev.params(_lib.SURFACE_SIX, 2)
    got = {}
    for fuse in (0, 1):
        with dev.tuning(entry_fuse=fuse):
            out, _ = dev.enhance_u8(frames, p)
            dev.check_status()
        got[fuse] = out.cpu().numpy()
    assert np.array_equal(got[0], got[1])
    for i in range(2):
        want = orc.enhance_u8(u8[i], 2)
        assert np.abs(got[1][i].astype(int) - want.astype(int)).max() <= 1
