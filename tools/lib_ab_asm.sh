#!/bin/bash
# tools/lib_ab_asm.sh OUT name1 name2 ...: like lib_ab.sh, printing the assembly-side kernel times
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
print("%-12s value %.2f  assemble %.4f  iq_update %.4f  convergence %.4f  spmv %.4f  its/newton %.2f" % (sys.argv[2], d["value"], k["assemble"]["avg_ms"], k["iq_update"]["avg_ms"], k["convergence"]["avg_ms"], k["spmv"]["avg_ms"], d["linear_iterations_per_newton"]), flush=True)
PY
done
