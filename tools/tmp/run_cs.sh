timeout 900 python -m pytest tests/test_gpu_adapter.py -x -q 2>&1 | tail -3
timeout 300 python tests/perf/client_step.py 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cs -o cs -- python3 $GRAFT_REPO_ROOT/tests/perf/client_step.py fused-only > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_cs -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220
