#!/bin/bash
# Per-launch durations of the filter kernels of one serialised benchmark frame (rocprofv3 kernel trace).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_lv
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lv -- python3 bench.py --no-cpu-baseline --no-overlap --no-lbvh --steps 1 --warmup 0 ${BENCH_EXTRA:-} > gpurun_out/prof_lv.log 2>&1 || { tail -5 gpurun_out/prof_lv.log; exit 1; }
f=$(find gpurun_out/prof_lv -name '*kernel_trace.csv' | head -1)
python3 tools/trace_durations.py "$f" k_filter | sed -n "1,${LINES_MAX:-12}p"
