#!/bin/bash
# tools/fused_ab.sh OUT: bench.py with and without --fused-reductions (one reduction per BiCGStab half iteration), alternating inside one GPU session
OUT=$1; mkdir -p $OUT
for rep in 1 2 3; do
for F in "" "--fused-reductions"; do
  N=plain; [ -n "$F" ] && N=fused
  python bench.py --full-line --steps 20 --warmup 5 --steady-after 0 --no-cpu-baseline --no-cpr-side-run $F > $OUT/${N}_$rep.json 2> $OUT/${N}_$rep.err || echo "$N failed"
  python - $OUT/${N}_$rep.json $N <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("%-6s value %.2f  spmv %.4f  ilu_apply %.4f  vector %.4f (x%d)  its/newton %.2f" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["vector"]["avg_ms"], k["vector"]["launches"], d["linear_iterations_per_newton"]), flush=True)
PY
done
done
