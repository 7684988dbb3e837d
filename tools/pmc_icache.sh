#!/bin/bash
# One PMC pass of tools/frame_loop.py: the instruction cache and scalar data cache of the sequencer
# (requests, hits, misses) beside the waves' cycles.  TAG names the output directory.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmci_${TAG:-x}
rm -rf $OUT
mkdir -p $OUT
CMD="python3 tools/frame_loop.py ${FRAMES:-2} ${SCENE_ARGS:-}"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; echo "pass $name done"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU && \
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
