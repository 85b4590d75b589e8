export OPMHIP_TUNING=1
for rep in 1 2; do
for L in 10 5 20 25 10; do
  timeout -k 10 200 python bench.py --full-line --steps 20 --warmup 5 --steady-after 0 --no-cpu-baseline --no-cpr-side-run --chain-length $L > gpurun_out/chain$L.json 2> gpurun_out/chain$L.err || echo "$L failed"
  python - gpurun_out/chain$L.json $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("chain %-3s value %.2f  spmv %.4f  ilu_apply %.4f  factor %.3f its/newton %.2f hp %s" % (sys.argv[2], d["value"], k["spmv"]["avg_ms"], k["ilu_apply"]["avg_ms"], k["ilu_factor"]["avg_ms"], d["linear_iterations_per_newton"], d.get("product_form",{}).get("half_product")), flush=True)
PY
done
done
