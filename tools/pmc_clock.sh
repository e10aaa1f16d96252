#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for be in table bitslice hybrid; do
  FLASHE_LIB_NAME=libflashe_hip_bitslice.so FLASHE_HYBRID_BS_PERMILLE=200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/$be -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prf-backend $be > $OUT/$be.log 2>&1
done
find $OUT -type f ! -name '*.csv' ! -name '*.log' -delete
