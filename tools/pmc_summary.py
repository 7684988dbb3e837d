#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes written by tools/pmc_passes.sh: per kernel, the sum of
every counter over all dispatches of one pass, plus the number of dispatches in a pass (DISPATCHES)."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r1"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(dict)
for path in sorted(glob.glob(f"{root}/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][path].add(r["Dispatch_Id"])
        dur[k][(path, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("kernel,counter,sum_over_dispatches")
for k in sorted(agg):
    if not k.startswith("k_"):
        continue
    for c in sorted(agg[k]):
        print(f"\"{k}\",{c},{agg[k][c]:.0f}")
    # launches of this kernel in one pass (every pass runs the same command)
    print(f"\"{k}\",DISPATCHES,{max(len(v) for v in calls[k].values())}")
