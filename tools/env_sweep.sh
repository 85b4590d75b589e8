#!/bin/bash
# tools/env_sweep.sh OUT "ENV1=a ENV2=b" "ENV1=c" ...: bench.py's kernel times under different environment settings
export OPMHIP_TUNING=1   # the library reads its measurement switches only under this master switch
OUT=$1; shift
mkdir -p $OUT
i=0
for E in "$@"; do
  i=$((i+1))
  env $E python bench.py --full-line --steps ${STEPS:-40} --warmup 5 --steady-after 0 --no-cpu-baseline > $OUT/e$i.json 2> $OUT/e$i.err || echo "$E failed"
  python - $OUT/e$i.json "$E" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-28s value %.2f  spmv %.4f  ilu_apply %.4f  factor %.4f  vector %.4f  asm %.4f  its/newton %.2f" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["ilu_factor"]["avg_ms"], k["vector"]["avg_ms"], k["assemble"]["avg_ms"], d["linear_iterations_per_newton"]), flush=True)
PY
done
