#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of one bench configuration; prints the top of the stats table.
# usage: tools/prof_cfg.sh <tag> <bench args...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" "$@" > "$OUT/bench.log" 2>&1 < /dev/null
echo "rc=$?" >> "$OUT/bench.log"
f=$(ls "$OUT"/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  cp "$f" "$OUT/kernel_stats.csv"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}")
PY
else
  tail -5 "$OUT/bench.log"
fi
find "$OUT/trace" -type f ! -name '*.csv' -delete 2>/dev/null
grep '^{"metric' "$OUT/bench.log" | cut -c1-400
