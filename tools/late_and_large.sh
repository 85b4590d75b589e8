cd $GRAFT_REPO_ROOT
for cfg in "ilu0 3" "cpr 3" "cpr 2"; do
  set -- $cfg
  timeout -k 10 500 python bench.py --full-line --steps 20 --warmup 5 --no-cpu-baseline --no-cpr-side-run --preconditioner $1 --cpr-reuse-setup $2 --steady-after 1500 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['steady_state']
print('$1 reuse $2: late window (from Newton %d): %.1f its/s, %.2f linear its per Newton, chopped %d' % (s['from_newton_iteration'], s['value'], s['linear_iterations_per_newton'], s['timesteps_chopped']))
"
done
timeout -k 10 500 python bench.py --full-line --n 200 --steps 20 --warmup 5 --no-cpu-baseline --no-cpr-side-run --steady-after 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('n=200 ilu0: %.2f its/s, %.2f lin/newton' % (d['value'], d['linear_iterations_per_newton']), {k:(v['avg_ms'], v.get('algorithmic_GBps')) for k,v in d['kernels'].items()}, d['stream_ceiling']['read_GBps'])
"
timeout -k 10 500 python bench.py --full-line --n 200 --steps 20 --warmup 5 --no-cpu-baseline --no-cpr-side-run --steady-after 0 --preconditioner cpr 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('n=200 cpr (ilu-1): %.2f its/s, %.2f lin/newton' % (d['value'], d['linear_iterations_per_newton']), d['kernels']['cpr_amg'])
"
