#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of an arbitrary python command; prints the stats table.
# usage: tools/prof_cmd.sh <tag> <script.py> <args...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/run.log" 2>&1 < /dev/null
f=$(ls "$OUT"/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  cp "$f" "$OUT/kernel_stats.csv"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.2f} min_us={float(r['MinNs'])/1e3:9.2f}")
PY
else tail -5 "$OUT/run.log"; fi
rm -rf "$OUT/trace"
