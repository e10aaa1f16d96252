timeout 900 python -m pytest tests -m gpu -x -q -k "sparse or span or config5 or bounds" 2>&1 | tail -3
for i in 1 2; do timeout 200 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_separate_launches'), d['phases_ms'])"; done
