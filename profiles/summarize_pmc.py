"""Fold rocprofv3 --pmc counter_collection.csv files (one per pass) into a per-kernel table.

usage: python profiles/summarize_pmc.py gpurun_out/pmc_<tag> [npx_total] [traffic.json nsteps]
(npx_total = pixels per step x steps run, warm-up included; with the last two arguments the per-kernel HBM bytes per
pixel and step are also written as JSON for bench.py's roofline.traffic)
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for wide streaming reads, MI355X_MICROARCH.md section HBM);
FETCH/WRITE_SIZE are in KB.
"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
npx = float(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
calls = collections.defaultdict(int)
for path in glob.glob(root + "/*/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("uwie::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (path, r["Dispatch_Id"])
        if key not in seen and "/sq/" in path:
            seen.add(key)
            dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            calls[name] += 1
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
print(f"{'kernel':42s} {'calls':>5s} {'ms':>8s} {'GBrd':>7s} {'GBwr':>7s} {'TB/s':>6s} {'B/px':>6s} " + " ".join(f"{c[3:]:>14s}" for c in cols))
for name in sorted(dur, key=lambda n: -dur[n]):
    a = acc[name]
    rd = 2 * a.get("FETCH_SIZE", 0) * 1024 / 1e9
    wr = a.get("WRITE_SIZE", 0) * 1024 / 1e9
    tbs = (rd + wr) / dur[name] if dur[name] else 0
    bpp = (rd + wr) * 1e9 / npx if npx else 0
    print(f"{name[:42]:42s} {calls[name]:5d} {dur[name]:8.3f} {rd:7.2f} {wr:7.2f} {tbs:6.2f} {bpp:6.1f} " +
          " ".join(f"{a.get(c, 0):14.3e}" for c in cols))

if len(sys.argv) > 4:
    import json

    nsteps = int(sys.argv[4])
    table = {}
    for name in dur:
        a = acc[name]
        tot = 2 * a.get("FETCH_SIZE", 0) * 1024 + a.get("WRITE_SIZE", 0) * 1024
        entry = {"hbm_bytes_per_px_per_step": round(tot / npx, 2), "launches_per_step": calls[name] // nsteps}
        table[name] = entry
        table.setdefault(re.sub(r"<.*", "", name), entry)  # also without template arguments
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md section HBM; profiles/collect_pmc.sh + summarize_pmc.py",
               "kernels": table}, open(sys.argv[3], "w"), indent=1)
