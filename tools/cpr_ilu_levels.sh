#!/bin/bash
# tools/cpr_ilu_levels.sh [levels ...]: bench.py --preconditioner cpr with 0 / 1 / 2 / 3 ILU0-smoothed AMG levels, alternating inside one GPU
# session; prints both windows, iterations per Newton iteration and the V-cycle's time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
for lv in ${@:-0 1 2}; do
  python bench.py --full-line --steps 20 --warmup 5 --no-cpu-baseline --preconditioner ${PREC:-cpr} --cpr-reuse-setup ${REUSE:-3} --cpr-amg-ilu-levels $lv 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['steady_state']
print('ilu_levels $lv: start-up %.1f its/s (%.2f lin/newton)  steady %.1f its/s (%.2f lin/newton)  V-cycle %.4f ms  ilu_apply %.4f spmv %.4f factor-scope %.3f' % (d['value'], d['linear_iterations_per_newton'], s['value'], s['linear_iterations_per_newton'], s['kernels']['cpr_amg']['avg_ms'], s['kernels']['ilu_apply']['avg_ms'], s['kernels']['spmv']['avg_ms'], s['kernels']['ilu_factor']['avg_ms']))
"
done
done
