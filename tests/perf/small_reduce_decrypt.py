#!/usr/bin/env python3
"""The reduce fused with the decrypt at b <= 64 (small_reduce_decrypt_kernel) against the two-launch form, HIP-event times of the
call alone (ten 1e7-element operands), alternated in one process; FLASHE_SMALL_REDUCE_PROBE=1 times the kernel without its AES rounds."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from flashe_amd.engine import Engine
n, C = 10_000_000, int(sys.argv[2])
b = int(sys.argv[1])
eng = Engine(bytes(range(32)), b)
cts = [eng.upload(np.random.default_rng(c).integers(0, 2 ** min(b, 63), n, dtype=np.uint64)) for c in range(C)]
out = eng.alloc_vec(n)
e0, e1 = eng.event(), eng.event()
run = lambda: eng.aggregate_decrypt_range_dev(0, [C], [0], n, 16, 0, n, cts, None, out)
for _ in range(40): run()
best = 1e9
for rep in range(6):
    eng.record(e0)
    for _ in range(10): run()
    eng.record(e1)
    best = min(best, eng.elapsed_ms(e0, e1) / 10)
print("%%.4f" %% best)
''' % ROOT
for b in [int(v) for v in sys.argv[1:]] or [20, 64]:
    for C in (10, 4, 16):
        row = []
        for name, env in (("two launches", {"FLASHE_SMALL_FUSED_REDUCE": "0"}), ("fused", {}), ("fused, no AES rounds (probe)", {"FLASHE_SMALL_REDUCE_PROBE": "1"})):
            r = subprocess.run([sys.executable, "-c", CODE, str(b), str(C)], env=dict(os.environ, FLASHE_LIB_NAME=os.environ.get("FLASHE_LIB_NAME", "libflashe_hip_tuning.so"), **env), capture_output=True, text=True)
            row.append(f"{name} {r.stdout.strip() or r.stderr[-300:]} ms")
        gb = (C + 1) * 10_000_000 * 8 / 1e9
        print(f"b={b} C={C} ({gb:.2f} GB): " + " | ".join(row), flush=True)
