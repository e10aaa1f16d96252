export FLASHE_LIB_NAME=libflashe_hip_tuning.so DECRYPT_ONLY=1
FLASHE_SPAN_PROBE=9 timeout 120 python tests/perf/sparse_phases.py 2>&1 | tail -10
