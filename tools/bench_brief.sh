#!/bin/bash
# Runs on the GPU box: ms_per_step + phases of the configurations named on the command line (default: 3 5 compact20 2 1), one line each.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
brief() { python "$REPO/bench.py" "$@" --no-cpu-baseline < /dev/null 2>/dev/null | grep '^{' | python -c '
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], "|", round(d["ms_per_step"], 4), "ms", {k: round(v, 4) for k, v in d.get("phases_ms", {}).items() if isinstance(v, float)},
      {k: round(v, 4) for k, v in d.items() if k.startswith("ms_per_step_") and v}, "frac", round(d["roofline"]["frac"], 4))' "$*"; }
for c in ${@:-3 5 compact20 2 1}; do
  case $c in
    compact*) brief --bits ${c#compact} --layout u32 ;;
    2) brief --no-e2e ;;
    *) brief --config $c ;;
  esac
done
