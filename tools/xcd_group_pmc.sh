#!/bin/bash
# time and FETCH_SIZE of the pipelined SpMV for several XCD group sizes
export OPMHIP_TUNING=1   # the library reads its measurement switches only under this master switch
R=$PWD
for G in 0 2 8 32; do
  export OPMHIP_XCD_GROUP=$G
  python bench.py --full-line --steps 12 --warmup 3 --steady-after 0 --no-cpu-baseline --no-cpr-side-run > gpurun_out/xg.json 2> gpurun_out/xg.err
  T=$(python - <<'PY'
import json
d=json.loads(open("gpurun_out/xg.json").read().strip().splitlines()[-1]); print(d["kernels"]["spmv"]["avg_ms"], d["kernels"]["ilu_apply"]["avg_ms"], round(d["value"],1))
PY
)
  (cd /tmp && export TMPDIR=/tmp && timeout -k 5 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/xg$G -o p -- python3 $R/bench.py --full-line --steps 4 --warmup 1 --no-cpu-baseline --steady-after 0 --no-cpr-side-run > /dev/null 2>&1)
  F=$(python3 - $G <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/xg%s/**/*counter_collection.csv"%sys.argv[1],recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for key in ("k_spmv_pipe","k_ilu_sweep_chain_LU","k_ilu_sweep_chain<"):
            if key in n: acc[key].append(float(r["Counter_Value"]))
print(" ".join("%s %.1f MB"%(k, 2*sum(v)/len(v)*1024/1e6) for k,v in sorted(acc.items())))
PY
)
  echo "G=$G  spmv/ilu ms, value: $T   fetched(x2): $F"
done
