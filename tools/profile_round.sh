#!/bin/bash
# rocprofv3 kernel trace + stats of the default benchmark command, then the PMC passes.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TAG=${1:-${TAG:-r3}}
mkdir -p gpurun_out
timeout -k 10 600 python3 -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_serial; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-config4 --no-config5 > gpurun_out/prof_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_$TAG.log; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_serial -- python3 bench.py --no-cpu-baseline --no-config4 --no-config5 --no-overlap --no-lbvh --frames-in-flight 1 > gpurun_out/prof_${TAG}_serial.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}_serial.log; exit 1; }
echo "kernel traces done (default command, and --no-overlap --frames-in-flight 1: one stream, one frame, every kernel alone on the chip)"
./tools/pmc_passes.sh
