#!/bin/bash
# PMC passes for the assembly kernels (run on the GPU box from the repo root): counters in their own runs, kernel-trace only.
set -e
OUT=${1:-gpurun_out/pmc_asm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $pass" >> $R/$OUT/progress.txt
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$OUT/p$i -o p -- python3 $R/tools/bench_assemble.py --reps 2 > $R/$OUT/p$i.log 2>&1 || echo "pass $i ($pass) failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_assemble" in k or "k_iq_update" in k or "k_newton" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
