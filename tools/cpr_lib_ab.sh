#!/bin/bash
# tools/cpr_lib_ab.sh OUT name1 name2 ...: bench.py --preconditioner cpr with build_variants/libopmhip_NAME.so in alternation inside one GPU session
OUT=$1; shift
mkdir -p $OUT
i=0
for V in "$@"; do
  i=$((i+1))
  export OPMHIP_LIB=build_variants/libopmhip_$V.so
  python bench.py --steps ${STEPS:-20} --warmup 5 --preconditioner ${PRECOND:-cpr} --no-cpu-baseline --no-cpr-side-run --detail $OUT/$i$V.detail.json > $OUT/$i$V.json 2> $OUT/$i$V.err || echo "$V failed"
  python - $OUT/$i$V.detail.json $V <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels"]
st = d.get("steady_state") or {}
print("%-10s value %.2f steady %.2f  cpr_amg %.4f  ilu_apply %.4f spmv %.4f vector %.4f  its/newton %.2f / %s" % (sys.argv[2], d["value"], st.get("value", 0.0), k["cpr_amg"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["spmv"]["avg_ms"], k["vector"]["avg_ms"], d["linear_iterations_per_newton"], st.get("linear_iterations_per_newton")), flush=True)
PY
done
