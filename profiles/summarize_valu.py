#!/usr/bin/env python3
"""Per-kernel VALU / LDS / SALU wave-instruction counts of one rocprofv3 PMC pass (profiles/quick.sh), per launch and per frame
pixel:  summarize_valu.py <dir> [bench args: --height H --width W --batch B]   (default workload 2160 x 3840 x 64).
lane-instructions per pixel = wave-instructions x 64 / pixels (the figure VERDICT.md quotes)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
args = sys.argv[2:]


def opt(name, default):
    return int(args[args.index(name) + 1]) if name in args else default


H, W, B = opt("--height", 2160), opt("--width", 3840), opt("--batch", 64)
npx = H * W * B
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uwie" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").replace("uwie::", "").replace("void ", "").split("(")[0][:56]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
# the bench runs warm-up + timed + recorded steps: normalise per step by the number of k_trans_init launches
steps = max(len(disp.get("k_trans_init", [])), 1)
print(f"# {H}x{W}x{B}, {steps} steps profiled; per step: ms (under the profiler), wave-instructions, lane-instructions per frame pixel")
tot = collections.Counter()
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    v = acc[k]
    valu, lds, salu = (v.get(c, 0) / steps for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"))
    tot["valu"] += valu; tot["lds"] += lds; tot["salu"] += salu; tot["ms"] += dur[k] / steps
    print(f"{k:58s} launches/step {len(disp[k]) / steps:5.1f}  ms {dur[k] / steps:7.3f}  VALU {valu:10.4g} ({valu * 64 / npx:6.1f}/px)  "
          f"LDS {lds:10.4g} ({lds * 64 / npx:5.1f}/px)  SALU {salu:10.4g}")
print(f"{'TOTAL':58s} ms {tot['ms']:7.3f}  VALU {tot['valu']:10.4g} ({tot['valu'] * 64 / npx:6.1f}/px)  LDS {tot['lds']:10.4g} ({tot['lds'] * 64 / npx:5.1f}/px)")
