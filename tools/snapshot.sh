#!/bin/bash
# tools/snapshot.sh NAME: the four files of a profiles/ snapshot, made in one GPU session and left under gpurun_out/NAME/
#   NAME_bench.json                un-profiled bench.py --steps 20 --warmup 5 (with the CPU baseline and the CPR side run): the ONE line the
#                                  driver parses; NAME_bench_detail.json: the full record behind it (per-window kernel scopes, reports)
#   NAME_bench_under_rocprof.json  the same command under rocprofv3 --kernel-trace --stats (no CPU baseline), + _detail.json
#   NAME_kernel_stats.csv          that run's per-kernel summary
#   NAME_pmc_traffic.json          FETCH_SIZE / WRITE_SIZE / TCC hit-miss per kernel, one --pmc pass per counter group
# (the raw traces stay in /tmp on the box: gpurun copies back 64 MiB at most).  Copy the four files to profiles/ afterwards.
set -e
N=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$N
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --detail $O/${N}_bench_detail.json > $O/${N}_bench.json 2> $O/bench.err
echo "bench done" > $O/progress.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$N
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail $O/${N}_bench_under_rocprof_detail.json > $O/${N}_bench_under_rocprof.json 2> $O/rocprof.err
cp "$(find /tmp/prof_$N -name '*kernel_stats.csv' | head -1)" $O/${N}_kernel_stats.csv
echo "rocprof done" >> $O/progress.txt
cd $R
bash tools/pmc_quick.sh gpurun_out/$N/pmc "bench.py --steps 6 --warmup 1 --no-cpu-baseline --steady-after 0 --no-cpr-side-run" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $O/pmc_summary.txt 2>&1
python3 tools/pmc_to_json.py $O/pmc $O/${N}_pmc_traffic.json > $O/pmc_fold.txt
rm -rf $O/pmc
echo "pmc done" >> $O/progress.txt
cat $O/${N}_bench_detail.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'steady', d['steady_state']['value'], 'cpr', d['cpr']['value'], d['cpr']['steady_state']['value'], 'cpu', d['cpu_baseline']['value'])
print('roofline', d['roofline'])
print({k:v['avg_ms'] for k,v in d['kernels'].items()})
print('stream', d['stream_ceiling']['read_GBps'])
"
cat $O/pmc_fold.txt
