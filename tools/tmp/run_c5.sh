timeout 200 python bench.py --config 5 --no-cpu-baseline > gpurun_out/c5_fused.json 2> gpurun_out/c5_fused.err
tail -c 600 gpurun_out/c5_fused.err
python - <<PY
import json
d=json.loads(open("gpurun_out/c5_fused.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("ms_per_step_separate_launches"), d["phases_ms"])
PY
timeout 200 python bench.py --config 5 --no-cpu-baseline --sparse-separate 2>&1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'])"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
