#!/bin/bash
# default bench line under different numbers of untimed settle rounds (clock ramp of the box)
for s in "$@"; do
  timeout 200 python bench.py --settle-rounds $s --no-cpu-baseline --no-e2e < /dev/null 2>/dev/null | S=$s python -c '
import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("settle", os.environ["S"], round(d["ms_per_step"], 4), d["phases_ms"])
'
done
