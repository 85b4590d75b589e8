#!/bin/bash
# PMC passes for the tile kernels (run on the GPU box from the repo root).  Counters in their own runs, kernel-trace only.
# usage: tools/pmc_run.sh <outdir> <reorder>
set -e
OUT=${1:-gpurun_out/pmc}; RO=${2:-line_coloring}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$OUT/$tag -o p -- python3 $R/tools/bench_kernels.py --n 100 --reorder $RO --reps 3 > $R/$OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 $R/tools/pmc_summary.py $R/$OUT > $R/$OUT/summary.txt
cat $R/$OUT/summary.txt
