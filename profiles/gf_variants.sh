#!/bin/bash
# Rebuilds k_guided_pipe.o with each extra flag set on the GPU box and times the variants (profiles/gf_bench.py).
#   bash profiles/gf_variants.sh "<gf_bench args>" "<flags A>" "<flags B>" ...      ("-" = no extra flags)
set -e
cd $GRAFT_REPO_ROOT/underwater_image_enhancement_amd/csrc
ARGS=$1; shift
for FL in "$@"; do
  [ "$FL" = "-" ] && FL=""
  echo "=== extra flags: [$FL]"
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=fast $FL -c k_guided_pipe.hip -o ../lib/obj/k_guided_pipe.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libuwie.so ../lib/obj/*.o
  (cd $GRAFT_REPO_ROOT && timeout -k 10 300 python profiles/gf_bench.py $ARGS 2>&1 | grep -v amdgpu.ids)
done
