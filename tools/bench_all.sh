#!/bin/bash
# Runs on the GPU box: every bench configuration once -> gpurun_out/bench_configs.jsonl (copied to profiles/ by hand).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/bench_configs.jsonl
mkdir -p "$REPO/gpurun_out"; : > "$OUT"
run() { echo "# bench.py $*" >> "$OUT"; timeout 600 python "$REPO/bench.py" "$@" < /dev/null 2>/dev/null | grep '^{' >> "$OUT"; }
run
run --schedule partial-agg --no-cpu-baseline --no-e2e
run --schedule auto --no-cpu-baseline --no-e2e
FLASHE_RCCL_SELF_SENDRECV=1 run --force-dist --no-cpu-baseline --no-e2e
run --config 1 --no-cpu-baseline
run --config 3 --no-cpu-baseline
run --config 3 --bits 23 --no-cpu-baseline
run --config 4 --no-cpu-baseline
run --config 4 --schedule partial-agg --no-cpu-baseline
run --config 5 --no-cpu-baseline
run --bits 64 --no-cpu-baseline --no-e2e
run --bits 20 --no-cpu-baseline --no-e2e
run --bits 20 --layout u32
run --bits 16 --no-cpu-baseline --no-e2e
run --bits 16 --layout u32
run --bits 24 --layout u32 --no-cpu-baseline
run --bits 32 --layout u32 --no-cpu-baseline
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"):
        print(l.strip()); continue
    d = json.loads(l)
    print("   ", round(d["ms_per_step"], 4), "ms", f'{d["value"]:.3e}', d.get("phases_ms"), "frac", round(d["roofline"]["frac"], 3))
PY
