#!/bin/bash
# Collects the PMC passes of the benchmark command, one rocprofv3 run per pass
# (counters of one hardware block must fit that block's slots).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${TAG:-r2}      # (remove the local copy of this directory before merging a new run into it: the summary adds up every file it finds)
rm -rf $OUT
mkdir -p $OUT
# two brute-force frames per pass: the timed one and the one of the kernel-timing leg
CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-lbvh --no-config4 --no-config5"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; echo "pass $name done"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run sq2 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run tcc TCC_HIT_sum TCC_MISS_sum && \
run grbm GRBM_GUI_ACTIVE && \
run mix SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT
