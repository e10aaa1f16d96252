#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/hyb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLASHE_LIB_NAME=libflashe_hip_bitslice.so FLASHE_HYBRID_BS_PERMILLE=300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prf-backend hybrid > $OUT/t.log 2>&1
find $OUT -type f ! -name '*.csv' ! -name '*.log' -delete
