#!/bin/bash
# tools/rest_sweep.sh OUT: the rest product's tile shape (OPMHIP_REST_ROWS / OPMHIP_REST_BLOCKS) and grid (OPMHIP_REST_WGS) under bench.py, alternating inside one GPU session
export OPMHIP_TUNING=1
OUT=$1; mkdir -p $OUT
run() {
  python bench.py --full-line --steps 20 --warmup 5 --steady-after 0 --no-cpu-baseline --no-cpr-side-run > $OUT/$1.json 2> $OUT/$1.err || echo "$1 failed"
  python - $OUT/$1.json "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-28s value %.2f  spmv %.4f  ilu_apply %.4f  factor %.4f  positions %s" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["ilu_factor"]["avg_ms"], d["product_form"]["rest_positions"]), flush=True)
PY
}
for rep in 1 2; do
  unset OPMHIP_REST_ROWS OPMHIP_REST_BLOCKS OPMHIP_REST_WGS; run default_$rep
  OPMHIP_REST_ROWS=32 run rows32_$rep
  unset OPMHIP_REST_ROWS; export OPMHIP_REST_ROWS=48; run rows48_$rep; unset OPMHIP_REST_ROWS
  export OPMHIP_REST_BLOCKS=160; run blocks160_$rep; unset OPMHIP_REST_BLOCKS
  export OPMHIP_REST_WGS=1536; run wgs1536_$rep; unset OPMHIP_REST_WGS
  export OPMHIP_REST_WGS=2560; run wgs2560_$rep; unset OPMHIP_REST_WGS
  export OPMHIP_REST_WGS=4096; run wgs4096_$rep; unset OPMHIP_REST_WGS
done
