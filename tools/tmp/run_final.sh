bash tools/bench_all.sh 2>&1 | tail -34
timeout 300 python tests/perf/client_step.py > gpurun_out/client_step.log 2>&1; cat gpurun_out/client_step.log
