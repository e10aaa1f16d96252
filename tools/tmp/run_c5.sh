timeout 900 python -m pytest tests -m gpu -x -q -k "sparse or span or config5 or bounds or bench_lines" 2>&1 | tail -5
timeout 200 python bench.py --config 5 --no-cpu-baseline > gpurun_out/c5_fused.json 2> gpurun_out/c5_fused.err
tail -c 600 gpurun_out/c5_fused.err
python - <<PY
import json
d=json.loads(open("gpurun_out/c5_fused.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("ms_per_step_separate_launches"), d["phases_ms"])
PY
timeout 600 python tests/perf/fuzz_sparse.py 60 7 2>&1 | tail -2
