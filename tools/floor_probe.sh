#!/bin/bash
# How long the scan launches of a frame last as the frame shrinks (teapot, depth 5, one frame at a time): the last
# frame's k_filter launches from a kernel trace, per frame size.  What does not shrink with the frame is the floor
# under a launch.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for size in "64 36" "256 144" "1024 576"; do
  d=gpurun_out/prof_floor_${size% *}
  rm -rf $d
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/frame_loop.py 4 scenes/teapot.scene $size 5 > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
  f=$(find $d -name '*kernel_trace.csv' | head -1)
  echo "== $size"
  python3 - "$f" <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_primary")]
frame = rows[starts[-1]:]
print("frame %.1f us, %d launches" % ((max(r[1] for r in frame) - frame[0][0]) / 1e3, len(frame)))
print(" ".join("%s:%.0f" % (r[2].split("(")[0].replace("void ", "").replace("k_filter", "F")[:12], (r[1] - r[0]) / 1e3) for r in frame if "k_filter" in r[2]))
print("others:", " ".join("%.0f" % ((r[1] - r[0]) / 1e3) for r in frame if "k_filter" not in r[2]))
PY
done
