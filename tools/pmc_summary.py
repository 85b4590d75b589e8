#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection csv files per kernel name and counter."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")) + glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        short = name.split("(")[0].replace("void ", "").replace("opmhip::", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(t in k for t in ("spmv", "ilu", "assemble", "iq_update", "bicg")):
        continue
    print(k)
    for cn, vals in sorted(acc[k].items()):
        print("    %-24s n=%-5d avg=%.4g" % (cn, len(vals), sum(vals) / len(vals)))
