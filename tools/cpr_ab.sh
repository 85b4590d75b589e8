#!/bin/bash
# tools/cpr_ab.sh OUT VAR v1 v2 ...: bench.py with the CPR preconditioner behind `value`, environment variable VAR set to each value
# ("-" = unset), alternating inside one gpurun call; prints Newton its/s of both windows and the V-cycle's time
export OPMHIP_TUNING=1   # the library reads its measurement switches only under this master switch
OUT=$1; VAR=$2; shift; shift
mkdir -p $OUT
for ROUND in 1 2; do
for V in "$@"; do
  if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
  python bench.py --steps ${STEPS:-20} --warmup 5 --preconditioner ${PRECOND:-cpr} --no-cpu-baseline --no-cpr-side-run --detail $OUT/$VAR$V.$ROUND.detail.json > $OUT/$VAR$V.$ROUND.json 2> $OUT/$VAR$V.$ROUND.err || echo "$V failed"
  python - $OUT/$VAR$V.$ROUND.detail.json $VAR=$V <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))   # the full record (the line on stdout carries one number per window)
k = d["kernels"]
st = d.get("steady_state") or {}
print("%-28s value %.2f steady %.2f  cpr_amg %.4f  ilu_apply %.4f spmv %.4f vector %.4f  factor+setup scope %.4f ms x %d  its/newton %.2f / %s" % (sys.argv[2], d["value"], st.get("value", 0.0), k["cpr_amg"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["spmv"]["avg_ms"], k["vector"]["avg_ms"], k["ilu_factor"]["avg_ms"], k["ilu_factor"]["launches"], d["linear_iterations_per_newton"], st.get("linear_iterations_per_newton")), flush=True)
PY
done
done
