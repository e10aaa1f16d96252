#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/bs2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for be in bitslice table; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/$be -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipeline-chunks 0 --prf-backend $be > $OUT/$be.log 2>&1
done
find $OUT -type f ! -name '*.csv' ! -name '*.log' ! -name '*.txt' -delete
