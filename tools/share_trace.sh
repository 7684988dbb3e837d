#!/bin/bash
# Kernel trace of what one rank of N renders (tools/sim_world.py N): per-kernel totals per frame,
# to tell GPU-busy time from gaps.  usage: share_trace.sh N
set -u
N=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_share
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_share -- python3 tools/sim_world.py $N > gpurun_out/prof_share.log 2>&1 || { tail -5 gpurun_out/prof_share.log; exit 1; }
tail -3 gpurun_out/prof_share.log
f=$(find gpurun_out/prof_share -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("all kernels: %.2f ms in %d launches (whole run)" % (tot / 1e6, calls))
for r in rows[:12]:
    print("%9.3f ms %6s calls  avg %8.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:60]))
PY
