#!/bin/bash
# Runs on the GPU box: samples rocm-smi (power, clocks) while the headline launch loops (tests/perf/ab_chain_libs.py on one or two builds).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
AB_REPS=${AB_REPS:-1500} python tests/perf/ab_chain_libs.py "$@" > /tmp/ab.log 2>&1 &
PID=$!
sleep 10
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -4
  sleep 1.5
done
wait $PID
tail -3 /tmp/ab.log
