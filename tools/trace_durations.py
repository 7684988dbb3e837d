"""Prints per-launch durations of the kernels whose name contains a pattern, from a rocprofv3
kernel-trace CSV (columns Kernel_Name, Start_Timestamp, End_Timestamp)."""
import csv, sys
path, pat = sys.argv[1], sys.argv[2]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:40]))
rows.sort()
for _, ms, name in rows:
    print(f"{ms:10.3f} ms  {name}")
