#!/bin/bash
# default bench line under two builds of the library (FLASHE_LIB_NAME), interleaved.  usage: ab_lib.sh <other .so name> [bench args]
OTHER=$1; shift
for i in $(seq 1 ${AB_ROUNDS:-3}); do
  for lib in libflashe_hip.so "$OTHER"; do
    FLASHE_LIB_NAME=$lib timeout 200 python bench.py --no-cpu-baseline --no-e2e "$@" < /dev/null 2>/dev/null | L=$lib python -c '
import sys, json, os
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(os.environ["L"], round(d["ms_per_step"], 4), d["phases_ms"])
'
  done
done
