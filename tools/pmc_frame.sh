#!/bin/bash
# rocprofv3 kernel trace and PMC passes of tools/frame_loop.py (frames on one stream, every kernel alone on the chip):
# one run per pass (counters of one hardware block must fit that block's slots).  TAG names the output directory.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcf_${TAG:-x}
mkdir -p $OUT
CMD="python3 tools/frame_loop.py ${FRAMES:-2} ${SCENE_ARGS:-}"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -3 $OUT/trace.log; exit 1; }
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; echo "pass $name done"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run sq2 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run tcc TCC_HIT_sum TCC_MISS_sum
