timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "c_example" --tb=short 2>&1 | tail -8
