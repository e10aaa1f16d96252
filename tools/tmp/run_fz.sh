FUZZ_VERBOSE=1 timeout 1200 python tests/perf/fuzz_round4.py 40 3 spenc 2>&1 | tail -6
