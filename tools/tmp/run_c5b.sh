time (timeout 600 python bench.py --config 5 > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err); tail -c 400 gpurun_out/c5_full.err
python - <<PY
import json
d=json.loads(open("gpurun_out/c5_full.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["cpu_baseline"], d["roofline"]["traffic"], d["roofline"]["frac"])
PY
