#!/bin/bash
# tools/lib_ab.sh OUT name1 name2 ...: bench.py kernel times with build_variants/libopmhip_NAME.so loaded (A/B inside one gpurun call)
OUT=$1; shift
mkdir -p $OUT
i=0
for V in "$@"; do
  i=$((i+1))
  export OPMHIP_LIB=build_variants/libopmhip_$V.so
  python bench.py --full-line --steps ${STEPS:-20} --warmup 5 --steady-after 0 --no-cpu-baseline --no-cpr-side-run > $OUT/$i$V.json 2> $OUT/$i$V.err || echo "$V failed"
  python - $OUT/$i$V.json $V <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-12s value %.2f  spmv %.4f  ilu_apply %.4f  factor %.4f  vector %.4f  its/newton %.2f" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["ilu_factor"]["avg_ms"], k["vector"]["avg_ms"], d["linear_iterations_per_newton"]), flush=True)
PY
done
