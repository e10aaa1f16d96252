for a in "--bits 20" "--bits 20 --layout u32" "--bits 64"; do timeout 300 python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', round(d['ms_per_step'],4), d.get('phases_ms'))"; done
timeout 1200 python -m pytest tests -m gpu -x -q -k "not fuzz and not multi_rank and not full_size" 2>&1 | tail -4
