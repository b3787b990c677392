#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters collected by profiles/pmc_sq.sh:  summarize_sq.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

root, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
calls = collections.Counter()
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uwie" not in k or flt not in k:
            continue
        k = k.split("uwie::(anonymous namespace)::", 1)[-1].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            if "/a/" in f:
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
                calls[k] += 1
for k in sorted(acc, key=lambda k: -dur[k]):
    v = acc[k]
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}: calls {calls[k]} ms {dur[k]:.3f}")
    print("   share of wave cycles: " + ", ".join(
        f"{c[3:]} {v[c] / wc:.2f}" for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
                                             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS") if c in v))
    print("   insts: " + ", ".join(f"{c[9:]} {v[c]:.4g}" for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD") if c in v)
          + f"; wave_cycles x4 {4 * v.get('SQ_WAVE_CYCLES', 0):.4g}; busy {v.get('SQ_BUSY_CYCLES', 0):.4g}; gui_active/8 {v.get('GRBM_GUI_ACTIVE', 0) / 8:.4g}"
          + f"; lds_idx_active {v.get('SQ_LDS_IDX_ACTIVE', 0):.4g}; bank_conflict {v.get('SQ_LDS_BANK_CONFLICT', 0):.4g}")
