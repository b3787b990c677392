#!/usr/bin/env python3
"""Static ISA evidence for every kernel of libuwie.so (round 3).

For each .hip file: cross-compile for gfx950 with the library's own flags plus `-Rpass-analysis=kernel-resource-usage`
and `--save-temps`, then report per kernel
  * VGPRs / AGPRs / SGPRs, scratch bytes per lane (spills or dynamically indexed private arrays), LDS bytes per block,
    occupancy in waves per SIMD (the compiler's figures);
  * the static instruction mix of the whole kernel and of its hot loop (the innermost loop -- a backward branch with no
    other backward branch inside -- with the most instructions), by class: VALU (f64 / f32 / int / cvt / transcendental),
    SALU, LDS (ds_*), VMEM (buffer_/global_/flat_), waitcnt, branch.

Runs without a GPU:  python profiles/isa_summary.py [--min-loop 24] [file.hip ...] > profiles/r03_isa_summary.txt
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "underwater_image_enhancement_amd", "csrc")
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
              "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-w"]
# per-file overrides, as in csrc/Makefile
OVERRIDE = {"k_guided_pipe.hip": {"-ffp-contract=off": "-ffp-contract=fast", "-std=c++17": "-std=c++20"}}

TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "valu_trans"
        if op.startswith("v_cvt"):
            return "valu_cvt"
        if "_f64" in op:
            return "valu_f64"
        if "_f32" in op or "_f16" in op:
            return "valu_f32"
        if op.startswith(("v_cmp", "v_cndmask")):
            return "valu_cmp_sel"
        return "valu_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:  # noqa: BLE001
        return names


def short(name):
    name = re.sub(r"uwie::\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def parse_asm(path):
    """{mangled kernel name: [(label or None, opcode), ...]}"""
    kernels, cur, name = {}, None, None
    with open(path) as f:
        for line in f:
            s = line.strip()
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", s)
            if m and cur is None:
                name, cur = m.group(1), []
                continue
            if cur is None:
                continue
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                cur.append((m.group(1), None, None))
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            parts = s.split()
            op = parts[0]
            target = parts[1] if op.startswith(("s_cbranch", "s_branch")) and len(parts) > 1 else None
            cur.append((None, op, target))
            if op == "s_endpgm":
                kernels[name] = cur
                cur = None
    return kernels


def mix(ins):
    c = {}
    for _, op, _ in ins:
        if op is None:
            continue
        k = classify(op)
        c[k] = c.get(k, 0) + 1
    return c


def hot_loop(ins, min_len):
    pos = {lab: i for i, (lab, _, _) in enumerate(ins) if lab}
    loops = []
    for i, (_, op, tgt) in enumerate(ins):
        if op and tgt in pos and pos[tgt] < i:
            loops.append((pos[tgt], i))
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]

    def largest(cands):
        best = None
        for lo, hi in cands:
            n = sum(1 for _, op, _ in ins[lo:hi + 1] if op)
            if n >= min_len and (best is None or n > best[0]):
                best = (n, lo, hi)
        return ins[best[1]:best[2] + 1] if best else None

    return largest(inner), largest(loops), len(loops)


def fmt_mix(c):
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    order = ["valu_f64", "valu_f32", "valu_int", "valu_cmp_sel", "valu_cvt", "valu_trans", "salu", "lds", "vmem", "scratch", "waitcnt", "branch"]
    body = " ".join(f"{k.replace('valu_', '')}={c[k]}" for k in order if c.get(k))
    return f"VALU={valu} [{body}]"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--min-loop", type=int, default=24)
    ap.add_argument("--hipcc", default="/opt/rocm/bin/hipcc")
    args = ap.parse_args()
    files = args.files or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print("# static ISA summary, gfx950 (profiles/isa_summary.py): resources as the compiler reports them,")
    print("# instruction mix of the whole kernel, of its largest loop and of its largest innermost loop (straight-line kernels have none)")
    for fn in files:
        src = os.path.join(CSRC, os.path.basename(fn))
        flags = [OVERRIDE.get(os.path.basename(fn), {}).get(f, f) for f in BASE_FLAGS]
        with tempfile.TemporaryDirectory() as tmp:
            cmd = [args.hipcc] + flags + ["-Rpass-analysis=kernel-resource-usage", "--save-temps", "-c", src, "-o", os.path.join(tmp, "x.o")]
            r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
            if r.returncode != 0:
                print(f"## {fn}: compile failed\n{r.stderr[-2000:]}")
                continue
            res, name = {}, None
            for line in r.stderr.splitlines():
                m = re.search(r"Function Name: (\S+)", line)
                if m:
                    name = m.group(1)
                    res[name] = {}
                    continue
                m = re.search(r"remark: \S+\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
                if m and name:
                    res[name][m.group(1)] = int(m.group(2))
            asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
            kernels = parse_asm(os.path.join(tmp, asm[0])) if asm else {}
        names = [n for n in res if n in kernels]
        pretty = dict(zip(names, demangle(names))) if names else {}
        print(f"\n## {os.path.basename(fn)}")
        for n in names:
            r_ = res[n]
            ins = kernels[n]
            total = sum(1 for _, op, _ in ins if op)
            loop, outer, nloops = hot_loop(ins, args.min_loop)
            print(f"{short(pretty[n])}")
            print(f"    VGPR {r_.get('VGPRs')}  AGPR {r_.get('AGPRs')}  SGPR {r_.get('TotalSGPRs')}  scratch {r_.get('ScratchSize [bytes/lane]')} B/lane"
                  f"  spills v{r_.get('VGPRs Spill')}/s{r_.get('SGPRs Spill')}  LDS {r_.get('LDS Size [bytes/block]')} B  occupancy {r_.get('Occupancy [waves/SIMD]')} waves/SIMD")
            print(f"    kernel   {total:5d} instr, {nloops} loops: {fmt_mix(mix(ins))}")
            if outer and outer is not loop and (loop is None or len(outer) != len(loop)):
                print(f"    main loop{sum(1 for _, op, _ in outer if op):5d} instr: {fmt_mix(mix(outer))}   (largest loop, inner loops included once)")
            if loop:
                print(f"    hot loop {sum(1 for _, op, _ in loop if op):5d} instr: {fmt_mix(mix(loop))}   (largest innermost loop)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
