#!/bin/bash
# Two PMC passes over tools/frame_loop.py that say what the scan kernels wait for: average latency of scalar loads,
# vector loads and instruction fetches (LEVEL / count), issue activity by unit.  TAG names the output directory.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcs_${TAG:-x}
mkdir -p $OUT
CMD="python3 tools/frame_loop.py ${FRAMES:-2} ${SCENE_ARGS:-}"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; echo "pass $name done"; }
run lat SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES && \
run act SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES
