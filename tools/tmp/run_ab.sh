timeout 600 python tests/perf/ab_two_libs.py libflashe_hip_ab.so 128 64 20 2>&1 | tail -4
for l in libflashe_hip.so libflashe_hip_ab.so; do FLASHE_LIB_NAME=$l timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-unchained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', round(d['ms_per_step'],4), d['phases_ms'], round(d['roofline']['frac'],4))"; done
for l in libflashe_hip.so libflashe_hip_ab.so libflashe_hip.so libflashe_hip_ab.so; do FLASHE_LIB_NAME=$l timeout 200 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', round(d['ms_per_step'],4), d['phases_ms'])"; done
