#!/bin/bash
# A/B of two library builds inside ONE gpurun call (boxes differ by ~5 %, so numbers from different calls do not compare):
#   make -C underwater_image_enhancement_amd/csrc OUT=../lib_b/libuwie.so OBJDIR=../lib_b/obj EXTRA=-D...   (here, before gpurun)
#   gpurun -- 'bash profiles/ab.sh [bench args]'      prints the per-kernel tables of A, B, A, B  (AB_ORDER="A B C A B C": a third
#   build in lib_c/)
R=$GRAFT_REPO_ROOT
run() {
  python3 - "$@" 2>&1 <<PY | grep -E "ms_per_step|^ +[0-9.]+ ms" | head -${AB_LINES:-8}
import sys, runpy, json
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--kernel-table", "--steps", "10", "--warmup", "2"] + sys.argv[2:]
sys.path.insert(0, "$R")
import underwater_image_enhancement_amd._lib as L
if sys.argv and "$1" == "B": pass
L.LIB_PATH = "$R/underwater_image_enhancement_amd/" + {"A": "lib", "B": "lib_b", "C": "lib_c"}["$1"] + "/libuwie.so"
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("$R/bench.py", run_name="__main__")
for l in buf.getvalue().splitlines():
    if l.startswith("{"):
        print("ms_per_step", json.loads(l)["ms_per_step"], "[$1]")
PY
}
for v in ${AB_ORDER:-A B A B}; do echo "== $v"; run $v "$@"; done
