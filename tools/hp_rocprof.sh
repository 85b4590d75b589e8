#!/bin/bash
# tools/hp_rocprof.sh OUT: per-kernel times of bench.py with the half-product form on and off (rocprofv3 --kernel-trace --stats, one run each,
# inside one GPU session); leaves OUT/hp{1,0}_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/$1
mkdir -p $O
export OPMHIP_TUNING=1
cd /tmp && export TMPDIR=/tmp
for V in 1 0; do
  rm -rf /tmp/prof_hp$V
  OPMHIP_HALF_PRODUCT=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hp$V -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --steady-after 0 --no-cpr-side-run --detail $O/hp${V}_detail.json > $O/hp${V}_bench.json 2> $O/hp${V}.err
  cp "$(find /tmp/prof_hp$V -name '*kernel_stats.csv' | head -1)" $O/hp${V}_kernel_stats.csv
  echo "hp=$V done" >> $O/progress.txt
  head -14 $O/hp${V}_kernel_stats.csv | cut -c1-200
done
