#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/pmc_quick.sh into per-kernel HBM traffic per launch.
usage: tools/pmc_to_json.py <pmc outdir> <out.json>
Correction prescribed by /opt/skills/guides/MI355X_MICROARCH.md ("HBM / rocprofv3" section): on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read -> doubled; WRITE_SIZE is exact; both are in KiB; Infinity-Cache hits are
counted (this is fabric-side traffic of the L2, an upper bound of the HBM bytes)."""
import collections, csv, glob, hashlib, json, os, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(out.items()):
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or k.startswith("__amd"):
        continue
    fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    res[k] = {"launches_sampled": len(d["FETCH_SIZE"]), "FETCH_SIZE_KiB_raw": fe, "WRITE_SIZE_KiB": wr,
              "traffic_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
    if "TCC_HIT_sum" in d:
        h, m = sum(d["TCC_HIT_sum"]) / len(d["TCC_HIT_sum"]), sum(d["TCC_MISS_sum"]) / len(d["TCC_MISS_sum"])
        res[k]["L2_hit_rate"] = h / (h + m)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hh = hashlib.sha256()
for f in ("solver.hip", "assemble.hip", "internal.hpp"):   # bench.py's kernel_source_hash(): ties the numbers to the build
    hh.update(open(os.path.join(ROOT, "opm-autodiff_amd", "csrc", f), "rb").read())
json.dump({"kernel_source_sha16": hh.hexdigest()[:16], "source": "rocprofv3 --kernel-trace --pmc, separate passes (tools/pmc_quick.sh) over bench.py --steps 6 --warmup 1",
           "correction": "FETCH_SIZE x 2 on gfx950, KiB -> bytes (MI355X_MICROARCH.md HBM/rocprofv3 section)", "kernels": res}, open(sys.argv[2], "w"), indent=1)
for k, v in res.items():
    print("%-46s %8.1f MB per launch" % (k, v["traffic_bytes_per_launch"] / 1e6))
