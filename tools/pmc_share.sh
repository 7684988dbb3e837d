#!/bin/bash
# Wave-cycle counters per kernel of one rank's share of an N-rank frame (tools/share_loop.py N 0 FRAMES, streams serialised)
# and of the whole frame (N = 1), to see what a share costs the chip beyond its fraction of the frame.  usage: pmc_share.sh N
set -u
N=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcs_$N
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 tools/share_loop.py $N 0 ${FRAMES:-4} 0 > $OUT/sq1.log 2>&1 || { tail -3 $OUT/sq1.log; exit 1; }
python3 tools/pmc_summary.py $OUT > $OUT.csv
python3 - $OUT.csv ${FRAMES:-4} <<'PY'
import csv, sys, collections
frames = int(sys.argv[2]) + 1          # (share_loop renders one frame before the timed ones)
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["kernel"]][r["counter"]] = float(r["sum_over_dispatches"])
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    # SQ_WAVE_CYCLES counts in units of four cycles; 256 CUs x 4 SIMDs x 4 wave slots at 2.4 GHz
    ms = v.get("SQ_WAVE_CYCLES", 0) * 4 / frames / (4096 * 2.4e9) * 1e3
    tot += ms
    print("%-34s %7.3f ms of four-wave slots per frame   waves %8.0f   launches %5.1f" % (k[:34], ms, v.get("SQ_WAVES", 0) / frames, v.get("DISPATCHES", 0) / frames))
print("sum %.3f ms" % tot)
PY
