"""One frame's launches on a time line, from a rocprofv3 kernel-trace CSV of tools/frame_loop.py run with the two streams on
(OVERLAP=1): per queue the kernels in order with their start relative to the frame's first kernel, their duration and the idle
time in front of each -- what a frame on its own waits for.  usage: frame_timeline.py trace.csv [frame_from_the_end]"""
import csv, sys
path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_primary")]
a = starts[-back]
b = starts[-back + 1] if back > 1 else len(rows)
frame = rows[a:b]
t0 = frame[0][0]
last_end = {}
busy = {}
print("frame: %d launches, %.3f ms from first start to last end" % (len(frame), (max(r[1] for r in frame) - t0) / 1e6))
for s, e, name, q in frame:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    print("q%-3s %9.1f us  +%7.1f us  gap %6.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name))
for q, v in busy.items():
    print("queue %s busy %.3f ms" % (q, v / 1e6))
