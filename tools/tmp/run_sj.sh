timeout 1500 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_parity.py -x -q -k "sparsif or sparse or wire" --tb=short 2>&1 | tail -5
timeout 600 python tests/perf/sparse_job_step.py > gpurun_out/sparse_job_step.log 2>&1; cat gpurun_out/sparse_job_step.log | tail -12
