export OPMHIP_TUNING=1   # the library reads its measurement switches only under this master switch
for L in 6 8 10 12 16; do
  python bench.py --full-line --steps 20 --warmup 5 --steady-after 0 --no-cpu-baseline --no-cpr-side-run --chain-length $L > gpurun_out/chain$L.json 2> gpurun_out/chain$L.err || echo "$L failed"
  python - gpurun_out/chain$L.json $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("chain %-3s value %.2f  spmv %.4f  ilu_apply %.4f  its/newton %.2f" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], d["linear_iterations_per_newton"]), flush=True)
PY
done
