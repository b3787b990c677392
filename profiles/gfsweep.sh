set -e
for cfg in 212 222 211 221 411 421; do for bands in 1 2; do
  echo "== CFG=$cfg BANDS=$bands"
  UWIE_GF_BANDS=$bands UWIE_GF_CFG=$cfg UWIE_BENCH_KERNELS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 2>&1 | grep -o "k_guided_wave=[0-9.]*\|\"ms_per_step\": [0-9.]*"
done; done
