#!/bin/bash
# tools/xcd_group_sweep.sh OUT G1 G2 ...: bench.py's kernel times for several OPMHIP_XCD_GROUP values (launch-schedule tuning)
export OPMHIP_TUNING=1   # the library reads its measurement switches only under this master switch
OUT=$1; shift
mkdir -p $OUT
for G in "$@"; do
  OPMHIP_XCD_GROUP=$G python bench.py --full-line --steps 20 --warmup 5 --steady-after 0 --no-cpu-baseline > $OUT/g$G.json 2> $OUT/g$G.err || echo "G=$G failed"
  python - $OUT/g$G.json $G <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("G=%-3s value %.2f  spmv %.4f ms  ilu_apply %.4f ms  its/newton %.2f" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], d["linear_iterations_per_newton"]), flush=True)
PY
done
