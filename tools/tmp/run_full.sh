timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err; python - <<PY
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["cpu_baseline"]["value"])
PY
