#!/bin/bash
# Runs on the GPU box: every differential fuzzer once with the seeds given (default 909 ...), last line of each -> stdout.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-909}
cd "$REPO"
timeout 500 python tests/perf/fuzz_chains.py 300 $S 2>&1 | tail -1
timeout 500 python tests/perf/fuzz_round3.py 16 $((S + 1)) 2>&1 | tail -1
timeout 600 python tests/perf/fuzz_round4.py 14 $((S + 2)) 2>&1 | tail -1
timeout 600 python tests/perf/fuzz_round5.py 10 $((S + 3)) 2>&1 | tail -1
timeout 400 python tests/perf/fuzz_dist.py 6 $((S + 4)) 2>&1 | tail -1
timeout 300 python tests/perf/fuzz_sparse.py 40 $((S + 5)) 2>&1 | tail -1
