#!/bin/bash
# tools/pmc_quick.sh <outdir> <script> [counter groups...]: one rocprofv3 --pmc pass per quoted group (kernel-trace only)
set -e
OUT=$1; shift
SCRIPT=$1; shift
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  echo "pass $i: $pass" >> $R/$OUT/progress.txt
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$OUT/p$i -o p -- python3 $R/$SCRIPT > $R/$OUT/p$i.log 2>&1 || echo "pass $i ($pass) failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if k.startswith("__amd") or "to_internal" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
