"""CPU-side checks of the C ABI: the library builds/loads and exports every symbol include/uwie.h declares."""
import ctypes
import os
import re

import pytest

import underwater_image_enhancement_amd as uw
from underwater_image_enhancement_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return uw.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "uwie.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uwie_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/uwie.h but not exported by libuwie.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_params_struct_matches_header_and_reference_defaults(lib):
    # six_stadigy.py:230-285
    want = {1: (0.3, 20, 0.5, 5, 98, 3.0, 1.5), 2: (0.5, 15, 0.5, 15, 95, 2.0, None), 3: (0.7, 10, 0.1, 20, 85, 0.0, None)}
    for k, (omega, ks, eps, lo, hi, clip, gamma) in want.items():
        p = uw.UwieParams()
        assert lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_SIX, k) == 0
        assert (round(p.omega, 6), p.gf_ksize, p.gf_eps, p.L_low, p.L_high, p.clip_limit) == (omega, ks, eps, lo, hi, clip)
        assert p.cast_correct == 1 and p.forced_cast == -1 and p.gray_shift == 15 and p.min_size == 1
        assert (p.tiles_x, p.tiles_y) == (8, 8)
        if gamma:
            assert p.gamma == gamma and p.apply_gamma == 1
    p = uw.UwieParams()
    assert lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_SIX, 3) == 0 and p.wb_percentile == 2
    assert lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_SIX, 4) == 0
    assert (p.clip_limit, p.L_low, p.L_high, p.wb_percentile, p.gamma) == (4.0, 10, 95, 3, 1.3)
    # enhancement_strategies.py in-code defaults (:356-372, :382-395, :428-441) and the fixed eps (:209)
    for name, (omega, ks, lo, hi) in {"strong_dehazing": (0.5, 15, 10, 95), "medium_dehazing": (0.6, 20, 15, 92),
                                      "light_enhancement": (0.4, 10, 15, 95)}.items():
        assert lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_DICT, _lib.DICT_STRATEGIES[name]) == 0
        assert (round(p.omega, 6), p.gf_ksize, p.L_low, p.L_high, p.gf_eps, p.cast_correct) == (omega, ks, lo, hi, 0.001, 0)
    assert lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_SIX, 7) != 0
    assert b"unknown strategy" in lib.uwie_last_error()
    assert lib.uwie_params_init(ctypes.byref(p), 9, 1) != 0


def test_workspace_sizes(lib):
    p = uw.UwieParams()
    lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_SIX, 2)
    small = lib.uwie_workspace_bytes(1, 480, 640, ctypes.byref(p))
    big = lib.uwie_workspace_bytes(64, 2160, 3840, ctypes.byref(p))
    assert 0 < small < big < 288 * 2**30  # fits one MI355X
    assert lib.uwie_workspace_bytes(0, 480, 640, ctypes.byref(p)) == 0
    assert lib.uwie_workspace_bytes(1, -1, 640, ctypes.byref(p)) == 0
    lib.uwie_params_init(ctypes.byref(p), _lib.SURFACE_SIX, 6)
    assert lib.uwie_workspace_bytes(1, 480, 640, ctypes.byref(p)) <= small


def test_null_arguments_are_rejected_without_a_gpu(lib):
    assert lib.uwie_enhance_u8(None, None, None, None, 1, 8, 8, None, None, 0, None) == -1
    assert lib.uwie_params_init(None, 0, 1) == -1
    assert lib.uwie_rgb2lab_u8(None, None, None, 0, None) == -1


def test_no_cpu_fallback_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(uw.UwieError):
        uw.get_device(0)
    import numpy as np

    with pytest.raises(uw.UwieError):
        uw.enhance(np.zeros((8, 8, 3), np.uint8))


def test_u8_over_255_identity():
    """k_guided_wave.hip computes g/255.0 as fma(fma(-q0, 255, g), 1/255, q0) with q0 = g * (1/255): three operations
    instead of a float64 division.  Checked here for every byte value with exact rational arithmetic (float(Fraction)
    rounds correctly, so each line below is one IEEE operation)."""
    from fractions import Fraction as F

    rcp = 1.0 / 255.0
    for g in range(256):
        q0 = float(g) * rcp
        r = float(F(g) - F(q0) * 255)
        q1 = float(F(r) * F(rcp) + F(q0))
        assert q1 == g / 255.0, g


def test_u8_over_255_identity_f32():
    """devutil.h px_norm_fast: the float32 version of the same identity (used by the quadtree chunk sums)."""
    from fractions import Fraction as F

    import numpy as np

    f32 = np.float32

    def rn(fr):  # exact rational -> nearest float32, ties to even
        y = f32(float(fr))
        cands = [np.nextafter(y, f32(-np.inf)), y, np.nextafter(y, f32(np.inf))]
        return min(cands, key=lambda c: (abs(F(float(c)) - fr), int(np.array([c], dtype=f32).view(np.uint32)[0]) & 1))

    rcp = f32(1.0) / f32(255.0)
    for u in range(256):
        x = f32(u)
        q0 = rn(F(float(x)) * F(float(rcp)))
        r = rn(F(float(x)) - F(float(q0)) * 255)
        q = rn(F(float(r)) * F(float(rcp)) + F(float(q0)))
        assert q == f32(u) / f32(255.0), u


def test_process_batch_log_rows_match_the_driver_columns():
    """Host bookkeeping of the batch driver (six_stadigy.py:369-500) with an injected compute: no GPU needed."""
    import numpy as np

    frames = np.zeros((2, 4, 5, 3), np.uint8)

    def fake(f):
        return {n: f.copy() for n, _ in uw.DRIVER_STRATEGIES}, ["greenish", "normal"]

    outs, rows, stats = uw.process_batch(frames, ["x.jpg", "y.jpg"], compute=fake)
    assert [r["strategy"] for r in rows[:6]] == ["strong_dehazing", "medium_dehazing", "light_dehazing",
                                                  "clahe_enhancement", "white_balance", "histogram_eq"]
    assert set(rows[0]) == {"filename", "image_type", "strategy", "strategy_desc", "status", "output_path", "processing_time"}
    assert rows[0]["image_type"] == "greenish" and rows[6]["filename"] == "y.jpg" and rows[6]["image_type"] == "normal"
    assert stats["image_types"] == {"greenish": 1, "bluish": 0, "normal": 1} and stats["successful_outputs"] == 12
    assert stats["processed_images"] == 2 and stats["total_images"] == 2
    with pytest.raises(ValueError):
        uw.process_batch(frames, ["only-one"], compute=fake)


def test_process_batch_failed_rows_follow_the_drivers_two_try_levels():
    """six_stadigy.py:424-480: a strategy that raises is a row with status 'failed', 'Error: <50 chars>' and 'N/A', counted in
    failed_outputs, while the image's other strategies still succeed; an image without a single success counts in
    failed_images (S6:484-488).  Injected computes, no GPU: the fused call fails for the batch and for image 1, whose
    strategy 3 then fails on its own; image 2 fails in every strategy."""
    import numpy as np

    frames = np.zeros((3, 4, 5, 3), np.uint8)
    frames[1] = 1
    frames[2] = 2

    def fused(f):
        if len(f) > 1 or f[0, 0, 0, 0] != 0:
            raise RuntimeError("fused path down")
        return {n: f.copy() for n, _ in uw.DRIVER_STRATEGIES}, ["bluish"] * len(f)

    def one(f, k):
        if f[0, 0, 0] == 2 or k == 3:
            raise RuntimeError("strategy %d broke: " % k + "x" * 80)
        return f + k

    outs, rows, stats = uw.process_batch(frames, ["a", "b", "c"], compute=fused, compute_one=one)
    assert len(rows) == 18 and [r["status"] for r in rows[:6]] == ["success"] * 6  # image 0: its own fused call worked
    b = rows[6:12]
    assert [r["status"] for r in b] == ["success", "success", "failed", "success", "success", "success"]
    assert b[2]["processing_time"] == "N/A" and b[2]["output_path"] == "Error: " + ("strategy 3 broke: " + "x" * 80)[:50]
    assert b[0]["processing_time"].endswith("s") and b[0]["output_path"] == ""
    assert all(r["status"] == "failed" for r in rows[12:])
    assert stats["processed_images"] == 2 and stats["failed_images"] == 1 and stats["total_outputs"] == 18
    assert stats["successful_outputs"] == 11 and stats["failed_outputs"] == 7
    assert outs["light_dehazing"][1] is None and np.array_equal(outs["medium_dehazing"][1], frames[1] + 2)
    assert outs["strong_dehazing"][2] is None and np.array_equal(outs["histogram_eq"][0], frames[0])
    # frames of different sizes run image by image through the fused call
    ragged = [np.zeros((4, 5, 3), np.uint8), np.zeros((6, 7, 3), np.uint8)]
    outs, rows, stats = uw.process_batch(ragged, compute=lambda f: ({n: f.copy() for n, _ in uw.DRIVER_STRATEGIES}, ["normal"] * len(f)))
    assert stats["successful_outputs"] == 12 and outs["white_balance"][1].shape == (6, 7, 3)


def test_no_entry_point_reads_the_environment():
    """Route selectors live in the context (uwie_set_tuning); UWIE_<NAME> variables only seed them, once, in uwie_create
    (round 2 read 25 variables per call).  The sources hold exactly one getenv call, in that seeding function."""
    src = os.path.join(ROOT, "underwater_image_enhancement_amd", "csrc")
    hits = []
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h", ".cpp")):
            text = re.sub(r"//.*", "", open(os.path.join(src, name)).read())
            hits += [(name, m.start()) for m in re.finditer(r"\bgetenv\s*\(", text)]
    assert [h[0] for h in hits] == ["api.hip"], hits
    api = open(os.path.join(src, "api.hip")).read()
    assert api.index("void tuning_from_env") < api.index("getenv(var)") < api.index("bool shape_ok")


def test_division_free_ab_to_xz_equals_the_c_expression():
    """k_fused.hip evaluates OpenCV's abToXZ_b entry without its three signed constant divisions (round 3):
    truncation = floor after adding 840 to a negative numerator, floor(n / 841) = (n * 5106980) >> 32 for n < 1.49e6, the
    cubic branch by 24-bit multiplies and shifts.  Same integers as the C expression for every argument the LAB2RGB path
    can produce (i = ify +- the a / b term, -8145 .. 26868)."""
    def c_div(a, b):
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b > 0) else -q

    def reference(i):
        if i <= 3390:
            return c_div(i * 108, 841) - c_div(c_div((1 << 14) * 16, 116) * 108, 841)
        return c_div(c_div(i * i, 1 << 14) * i, 1 << 14)

    def device(i):
        t = i * 108
        tp = (t + (840 if t < 0 else 0) + 841 * 1100) & 0xffffffff
        lin = ((tp * 5106980) >> 32) - (1100 + 290)
        u = i & 0xffffff  # v_mul_u32_u24 takes the low 24 bits of its operands
        sq = ((u * u) & 0xffffffff) >> 14
        cub = (((sq & 0xffffff) * u) & 0xffffffff) >> 14
        return lin if i <= 3390 else cub

    assert all(reference(i) == device(i) for i in range(-8145, 26869))


def test_no_v_ashr_pk_in_the_byte_saturating_kernels(tmp_path):
    """Round 3 compiler finding (Exactness): `saturate_cast<uchar>(x >> n)` of two adjacent values is selected as one
    v_ashr_pk_u8_i32 on gfx950, and its result did not match on the hardware.  The kernels that saturate shifted integers to
    bytes (RGB2LAB / LAB2RGB / CLAHE) write clamp-then-shift; this compiles them to assembly and checks that no packed
    shift-and-saturate instruction is left (hipcc cross-compiles without a GPU)."""
    import subprocess

    src = os.path.join(ROOT, "underwater_image_enhancement_amd", "csrc")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-fno-fast-math", "-w", "-S", "--cuda-device-only"]
    for name in ("k_fused.hip", "k_clahe.hip", "k_codes.hip"):
        out = tmp_path / (name + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(src, name), "-o", str(out)])
        text = out.read_text()
        assert "s_endpgm" in text
        assert "v_ashr_pk" not in text, f"{name}: v_ashr_pk_* selected"


def test_cast_detection_kernel_assumptions():
    """The arithmetic facts k_entry.hip's round-3 kernels lean on, restated with NumPy (no GPU):
    * R_e(v) = RN(x_v / ulp_e) < 2^26 for every byte value and binade e >= -2, so R = hi * 2^13 + lo with 13-bit halves and,
      with at most 16384 pixels per chunk, both partial sums of k_chunk_ulps stay below 2^27 (v_dot2_u32_u16 operands are
      16 bits, its accumulator 32);
    * a byte value ties (x_v / ulp_e = n + 1/2) in at most ONE binade (CastTables::tiebin);
    * a lane column of k_chunk_hist receives at most 512 + 1 pixels of a chunk: 10-bit fields cannot overflow;
    * 16 table entries of k_cast_resolve (count below 2^26, tie flag at bit 40) cannot carry into each other."""
    import numpy as np

    x = (np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float64)
    ties = np.zeros((34, 256), bool)
    for ei, e in enumerate(range(-2, 32)):
        y = x / 2.0 ** (e - 23)  # exact: a power of two
        r = np.floor(y)
        frac = y - r
        ties[ei] = frac == 0.5
        r = np.where(frac > 0.5, r + 1, r)
        assert r.max() < 2 ** 26
        lo, hi = r.astype(np.int64) & 0x1FFF, r.astype(np.int64) >> 13
        assert hi.max() < 2 ** 13 and 16384 * max(lo.max(), hi.max()) < 2 ** 27
        assert np.array_equal((hi << 13) + lo, r.astype(np.int64))
    assert ties.sum(axis=0).max() <= 1 and not ties[:, 0].any()
    assert 16384 // 32 + 3 < 1024
    assert 16 * (2 ** 26) < 2 ** 40 and 16 < 2 ** (64 - 40)


def test_gray_quantisation_identities():
    """k_q_hist writes the 8-bit gray plane from the bytes themselves: (x * 255).astype(u8) of the normalised value is
    the byte again, and of the attenuated value (x * 0.85) it is 17 u // 20 -- NumPy's float32 arithmetic, all 256 bytes."""
    import numpy as np

    u = np.arange(256)
    x = u.astype(np.float32) / np.float32(255.0)
    assert np.array_equal((x * np.float32(255.0)).astype(np.uint8), u)
    assert np.array_equal(((x * np.float32(0.85)) * np.float32(255.0)).astype(np.uint8), u * 17 // 20)


def test_no_vgpr_spill_ahead_of_an_exec_restore():
    """profiles/isa_lint.py over the device assembly the build keeps (lib/obj/*-gfx950.s): the miscompile behind round 3's
    255-LSB build of k_stretch_lab_lut<1, 256> -- VGPR spill stores at the head of a join block, ahead of the s_or_b64 that
    restores EXEC, so lanes that sat the branch out lose their threadIdx.x (profiles/r04_spill_miscompile.txt) -- fails the
    build here instead of on the GPU.  The lint itself is checked on the two excerpts kept from that investigation."""
    import glob
    import importlib.util
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(root, "profiles", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    bad = """k_demo:
	s_and_saveexec_b64 s[22:23], s[0:1]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v1, v2
.LBB0_2:
	scratch_store_dwordx2 off, v[76:77], off offset:12 ; 8-byte Folded Spill
	s_or_b64 exec, exec, s[22:23]
	s_endpgm
"""
    good = bad.replace("\tscratch_store_dwordx2 off, v[76:77], off offset:12 ; 8-byte Folded Spill\n\ts_or_b64 exec, exec, s[22:23]\n",
                       "\ts_or_b64 exec, exec, s[22:23]\n\tscratch_store_dwordx2 off, v[76:77], off offset:12 ; 8-byte Folded Spill\n")
    assert good != bad
    with tempfile.TemporaryDirectory() as d:
        for name, text, n in (("bad.s", bad, 1), ("good.s", good, 0)):
            path = os.path.join(d, name)
            with open(path, "w") as f:
                f.write(text)
            assert len(lint.lint(path)) == n, name
    files = sorted(glob.glob(os.path.join(os.path.dirname(_lib.LIB_PATH), "obj", "*gfx950.s")))
    if not files:  # a library built before the Makefile kept the assembly
        _lib.build(force=True)
        files = sorted(glob.glob(os.path.join(os.path.dirname(_lib.LIB_PATH), "obj", "*gfx950.s")))
    assert len(files) >= 17, "the build keeps one device assembly file per .hip source"
    findings = [f for p in files for f in lint.lint(p)]
    assert not findings, "\n".join(findings)
