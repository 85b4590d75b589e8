#!/bin/bash
# tools/env_ab.sh OUT VAR v1 v2 ...: bench.py kernel times with the environment variable VAR set to each value (A/B inside one gpurun call)
export OPMHIP_TUNING=1   # the library reads its measurement switches only under this master switch
OUT=$1; VAR=$2; shift; shift
mkdir -p $OUT
for V in "$@"; do
  export $VAR=$V
  python bench.py --full-line --steps ${STEPS:-20} --warmup 5 --steady-after 0 --no-cpu-baseline --no-cpr-side-run > $OUT/$VAR$V.json 2> $OUT/$VAR$V.err || echo "$V failed"
  python - $OUT/$VAR$V.json $VAR=$V <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-22s value %.2f  spmv %.4f  ilu_apply %.4f  factor %.4f  vector %.4f  its/newton %.2f" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["ilu_factor"]["avg_ms"], k["vector"]["avg_ms"], d["linear_iterations_per_newton"]), flush=True)
PY
done
