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
