"""Device-side self checks: a defect inside the library must surface as an error code, not as a GPU memory access fault.

VERDICT r03 item 2 asked for the cause of round 3's two memory access faults (gpurun_out/r3_hazy5_kernels.txt,
r3_uni8_kernels.txt) and for one test per cause.  The cause: while the weak-only Canny counting
was being changed, k_canny_gradnms stopped writing label[] for the tile-local ROOT of a component when that root is an
interior pixel (only border candidates got a label and a list slot); k_canny_mark / k_canny_emit walked from a border member
to the root's pixel and on through whatever an earlier quadtree level had left in the label plane -- an index outside the
frame.  Tuning canny_fault_inject puts that defect back; the walkers now validate every label before they follow it.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uw():
    import underwater_image_enhancement_amd as uw

    return uw


def test_canny_walkers_report_a_stale_label_instead_of_faulting(uw):
    import torch

    from underwater_image_enhancement_amd import _lib

    dev = uw.get_device()
    rng = np.random.default_rng(31)
    # the hazy range of SURVEY 8(d): dense weak candidates whose components cross the 32 x 64 tile seams
    u8 = np.floor(255 * (rng.random((2, 512, 768, 3)) * 0.7 + 0.15)).astype(np.uint8)
    frames = dev.tensor(u8)
    p = dev.params(_lib.SURFACE_SIX, 2)
    want, _ = dev.enhance_u8(frames, p)
    dev.check_status()  # a correct launch never sets the word
    want = want.cpu().numpy()
    # label[] is never cleared: what a walker finds at an unwritten root is whatever the workspace held.  Make that
    # "whatever" an index far outside the frame, as a stale label of a larger frame would be.
    dev.workspace_for(2, 512, 768, p).fill_(0x7f)
    with dev.tuning(canny_fault_inject=1):
        dev.enhance_u8(frames, p)  # must return: every dereferenced index is validated
        with pytest.raises(_lib.UwieError, match="Canny"):
            dev.check_status()
    dev.check_status()  # reading the word cleared it
    torch.cuda.synchronize()
    got, _ = dev.enhance_u8(frames, p)
    dev.check_status()
    assert np.array_equal(got.cpu().numpy(), want)  # the poisoned workspace itself changes nothing on the correct path


def test_tile_lut_kernel_with_idle_lanes_in_its_prefetch_branch(uw):
    """The second cause VERDICT r03 item 2 asked for: round 3's 255-LSB build of k_stretch_lab_lut<1, 256> was a MISCOMPILE, not
    undefined behaviour in the source -- over its 102-register budget the allocator spilled threadIdx.x at the head of the join
    block of `if (fastpath && tid < total) prefetch`, ahead of the s_or_b64 that restores EXEC, so lanes with tid >= total
    reloaded garbage and indexed the tile LUT with it (profiles/r04_spill_miscompile.txt; profiles/isa_lint.py now fails such
    a build on the CPU).  This is the GPU-side sentinel: tiles with fewer pixel groups than the block has threads (idle lanes
    in that branch: total = 24, 48, 252), exactly as many (256) and more, strategies 1 and 2, against the oracle."""
    from oracle import uwie_oracle as orc

    rng = np.random.default_rng(404)
    for H, W in ((61, 83), (96, 128), (144, 448), (128, 512), (256, 640)):
        u8 = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for k in (1, 2):
            d = np.abs(uw.enhance(u8, strategy=k).astype(int) - orc.enhance_u8(u8, k).astype(int))
            assert d.max() <= (1 if k == 1 else 0), f"{H}x{W} strategy {k}: {d.max()} LSB, {np.count_nonzero(d)} bytes differ"


def test_quadtree_scores_lie_inside_their_histogram_intervals(uw):
    """the histogram-decided quadtree levels rest on hand-derived rounding bounds ((nch + 35) u, (nch + 38) u, 8 u of
    slack, times 1.25); route-equality tests only show that no decision differed on the frames tried.  Tuning q_hist = 3 takes
    the histograms AND runs the reference-order kernels, and the device itself checks, for every launched level and every level
    of k_q_tail, that each quadrant's reference-order score lies inside its interval and that a decision the intervals allowed
    is the reference argmax (UWIE_STATUS_QTREE_BOUNDS otherwise).  Adversarial inputs: near-tie quadrants, a saturated frame,
    black, flat, two-level, noise with many 8192-element chunks per quadrant (a 4K frame), every cast kind."""
    import torch

    dev = uw.get_device()
    rng = np.random.default_rng(314)
    H, W = 460, 700
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8), np.full((H, W, 3), 255, np.uint8), np.zeros((H, W, 3), np.uint8),
              np.full((H, W, 3), 97, np.uint8), np.clip(rng.normal(128, 3, (H, W, 3)), 0, 255).astype(np.uint8),
              np.clip(rng.normal(250, 6, (H, W, 3)), 0, 255).astype(np.uint8)]
    tie = np.full((H, W, 3), 120, np.uint8)  # four quadrants that differ in one pixel each
    for i, (y, x) in enumerate(((10, 10), (10, 600), (400, 10), (400, 600))):
        tie[y, x] = 121 + i
    frames.append(tie)
    batch = dev.tensor(np.stack(frames))
    for kind_id in (0, 1, 2):
        kk = torch.full((len(frames),), kind_id, dtype=torch.int32, device=dev.torch_device)
        with dev.tuning(q_hist=1):
            want = dev.atmospheric_light(batch, kk).cpu().numpy()
        with dev.tuning(q_hist=3):
            got = dev.atmospheric_light(batch, kk).cpu().numpy()
            dev.check_status()  # raises on UWIE_STATUS_QTREE_BOUNDS
        assert np.array_equal(got, want), kind_id
    big = rng.integers(0, 256, (2, 2160, 3840, 3), dtype=np.uint8)  # 253 chunks per level-0 quadrant
    big[1] = np.clip(big[1].astype(int) // 8 + 200, 0, 255).astype(np.uint8)
    with dev.tuning(q_hist=3):
        a3 = dev.atmospheric_light(dev.tensor(big)).cpu().numpy()
        dev.check_status()
    assert np.array_equal(a3, dev.atmospheric_light(dev.tensor(big)).cpu().numpy())
