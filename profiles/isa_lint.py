#!/usr/bin/env python3
"""Static check of the device assembly the build keeps (underwater_image_enhancement_amd/lib/obj/*-gfx950.s): no VGPR spill
store or reload may sit between the head of a basic block and the s_or_b64 that restores EXEC there.

Why: `if (tid < total) { prefetch }` in k_stretch_lab_lut<1, 256> ends in a join
block whose first instruction is `s_or_b64 exec, exec, s[22:23]` -- the lanes with tid >= total come back there.  In the
round-3 build that went over its 102-register budget, the register allocator (ROCm 7.2, clang 22) put three spill stores
-- among them the register that holds threadIdx.x -- at the head of that block, AHEAD of the s_or_b64: the lanes that sat
the branch out never wrote their copy, the reload further down (full EXEC) handed them whatever the scratch slot held, and
the tile's LUT was indexed with it: 255-LSB errors on every frame whose tiles have fewer than 256 pixel groups.  The source
was correct; the same source unspilled is correct.  This lint turns that class of miscompile into a build-time failure.

usage: isa_lint.py [file.s ...]   (default: every *.s under lib/obj); exit status 1 and one line per finding.
"""
import glob
import os
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:|[A-Za-z_][\w$.]*:)")
SPILL = re.compile(r"^\s*(scratch_(store|load)|buffer_(store|load))\w*\s.*;\s*\d+-byte Folded (Spill|Reload)")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
EXEC_OTHER = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_(and|or|xor|andn2)_saveexec_b64")
BRANCH = re.compile(r"^\s*s_(c?branch|endpgm|setpc)")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")


def lint(path):
    findings = []
    func = "?"
    pending = []  # spill / reload instructions seen since the head of the current block, no EXEC write in between
    with open(path) as f:
        for no, line in enumerate(f, 1):
            m = FUNC.match(line)
            if m and not line.startswith(".L"):
                func = m.group(1)
            if LABEL.match(line) or BRANCH.match(line):
                pending = []
                continue
            if SPILL.match(line):
                pending.append((no, line.strip()))
                continue
            if EXEC_RESTORE.match(line):
                for pno, pline in pending:
                    findings.append(f"{os.path.basename(path)}:{pno}: {func}: `{pline}` executes under the partial EXEC mask "
                                    f"that line {no} (`{line.strip()}`) widens")
                pending = []
                continue
            if EXEC_OTHER.match(line):
                pending = []
    return findings


def main(argv):
    files = argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                       "underwater_image_enhancement_amd", "lib", "obj", "*gfx950.s")))
    if not files:
        print("isa_lint: no assembly files (build first: make -C underwater_image_enhancement_amd/csrc)")
        return 2
    bad = []
    for p in files:
        bad += lint(p)
    for b in bad:
        print(b)
    print(f"isa_lint: {len(files)} files, {len(bad)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
