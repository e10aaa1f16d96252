#!/bin/bash
# Runs on the GPU box: average shader clock of every dispatch of a python command = GRBM_GUI_ACTIVE / 8 XCDs / duration.
# usage: tools/prof_clock.sh <tag> <script.py> <args...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/run.log" 2>&1 < /dev/null
f=$(ls "$OUT"/trace/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
for r in rows:
    if r.get("Counter_Name") != "GRBM_GUI_ACTIVE" or "chain" not in r["Kernel_Name"]:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
    cyc = float(r["Counter_Value"]) / 8
    print(f"{r['Kernel_Name'][:40]:40s} dur_us={dur / 1e3:9.1f} cycles={cyc:12.0f} clock_GHz={cyc / dur if dur else 0:.3f}")
PY
else tail -5 "$OUT/run.log"; fi
tail -8 "$OUT/run.log"
rm -rf "$OUT/trace"
