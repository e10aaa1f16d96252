#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench, then two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md's HBM section prescribes.
# Raw output -> gpurun_out/prof/<tag>/ ; summarise with tools/summarize_profile.py.
set -u
TAG=${1:-r04}
STEPS=${2:-20}
shift; shift                    # further arguments go to bench.py (e.g. --config 5)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-e2e --no-unchained "$@" > "$OUT/trace.log" 2>&1 < /dev/null
echo "trace rc=$?" >> "$OUT/trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-unchained "$@" > "$OUT/pmc_fetch.log" 2>&1 < /dev/null
echo "fetch rc=$?" >> "$OUT/pmc_fetch.log"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-unchained "$@" > "$OUT/pmc_write.log" 2>&1 < /dev/null
echo "write rc=$?" >> "$OUT/pmc_write.log"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-unchained "$@" > "$OUT/pmc_sq.log" 2>&1 < /dev/null
echo "sq rc=$?" >> "$OUT/pmc_sq.log"
# keep only the CSVs (the merge-back limit is 64 MiB)
find "$OUT" -type f ! -name '*.csv' ! -name '*.log' -delete
du -sh "$OUT"
