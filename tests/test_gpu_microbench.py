"""Hardware facts the library's code generation rests on, measured by small stand-alone HIP programs (profiles/microbench/)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_v_ashr_pk_u8_i32_writes_the_low_half_only():
    """DESIGN.md section 4, "A compiler finding": gfx950's v_ashr_pk_u8_i32 computes sat_u8(x0 >> (s & 31)) | sat_u8(x1 >> (s & 31)) << 8
    into the LOW 16 bits of its destination and leaves the high 16 as they were, while this toolchain stores the register as if
    the instruction had zero-extended it (the round-3 wrong a / b bytes of RGB2LAB).  The library therefore keeps the pattern that
    selects it out of its sources (tests/test_cabi.py::test_no_v_ashr_pk_in_the_byte_saturating_kernels); this test pins the two
    hardware facts.  What the compiler does with the C++ pattern is printed, not asserted: a fixed toolchain must not fail here."""
    exe = os.path.join(ROOT, "profiles", "microbench", "ashr_pk")
    if not os.path.exists(exe):
        pytest.skip("profiles/microbench/ashr_pk is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    print(out)
    m = re.search(r"x0 -> byte 0\)\): (\d+) of (\d+) differ", out)
    assert m and int(m.group(1)) == 0, out
    m = re.search(r"high 16 bits of the destination: kept in (\d+), zero in (\d+) of (\d+)", out)
    assert m and int(m.group(1)) == int(m.group(3)), out


@pytest.mark.gpu
def test_two_8_byte_buffer_stores_have_no_store_data_hazard():
    """on gfx950 a VALU write to the data registers of a 16-byte buffer store issued right behind the store
    can reach memory (dword 0 of the store), with soffset in an SGPR as well as with an immediate one -- LLVM's hazard recognizer
    only covers the second form, which is how compiler-generated code of round 2 stored wrong transmissions.  k_guided_pipe /
    k_guided_split therefore store 8 bytes at a time.  Pinned here: the 8-byte form is clean, and one wait state repairs the
    SGPR form (what the 16-byte experiment of round 4 relied on).  The hazard itself is printed, not asserted (it is a race)."""
    exe = os.path.join(ROOT, "profiles", "microbench", "store_hazard")
    if not os.path.exists(exe):
        pytest.skip("profiles/microbench/store_hazard is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    print(out)
    rows = {m.group(1).strip(): (int(m.group(2)), int(m.group(3))) for m in
            re.finditer(r"^(.*?)\s*: (\d+) of \d+ dwords hold the poison .*?, (\d+) other mismatches", out, re.M)}
    assert len(rows) == 6, out
    assert all(other == 0 for _, other in rows.values()), out
    assert rows["two 8-byte stores, soffset in an SGPR, v_mov behind"][0] == 0, out
    assert rows["16-byte store, soffset in an SGPR, s_nop 0 between"][0] == 0, out
