#!/bin/bash
# tools/variant_sweep.sh OUT name1 name2 ...: bench.py's kernel times for libopmhip variants under build_variants/ ("default" = the in-tree library)
OUT=$1; shift
mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = default ]; then unset OPMHIP_LIB; else export OPMHIP_LIB=$PWD/build_variants/libopmhip_$V.so; fi
  python bench.py --full-line --steps ${STEPS:-40} --warmup 5 --steady-after 0 --no-cpu-baseline > $OUT/$V.json 2> $OUT/$V.err || echo "$V failed"
  python - $OUT/$V.json $V <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-12s value %.2f  spmv %.4f  ilu_apply %.4f  factor %.4f  vector %.4f  asm %.4f  its/newton %.2f  stream %.0f GB/s" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["ilu_factor"]["avg_ms"], k["vector"]["avg_ms"], k["assemble"]["avg_ms"], d["linear_iterations_per_newton"], d["stream_ceiling"]["read_GBps"]), flush=True)
PY
done
