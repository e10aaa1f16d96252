#!/usr/bin/env python3
"""Round 6: the chained launch on short vectors in HALF tiles (128 counters, one pair of counters per lane) against QUARTER tiles
(64 counters, one block per lane, two streams per step), and the pieces-per-chain knob in quarter mode.  Tuning build
(FLASHE_CHAIN_TUNE knobs are read per launch).  usage: quarter_sweep.py [bits]"""
import os
import sys

os.environ.setdefault("FLASHE_LIB_NAME", "libflashe_hip_tuning.so")
os.environ["FLASHE_CHAIN_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = Engine(bytes(range(32)), b)
e0, e1 = eng.event(), eng.event()


def timeit(fn, inner=20, reps=5):
    for _ in range(20):
        fn()
    best = 1e9
    for _ in range(reps):
        eng.record(e0)
        for _ in range(inner):
            fn()
        eng.record(e1)
        best = min(best, eng.elapsed_ms(e0, e1) / inner)
    return best * 1e3


for n, C in ((61_706, 100), (61_706, 10), (250_000, 100), (250_000, 10), (100_003, 30), (20_000, 40), (5_000, 3), (1_000, 100), (1_000_000, 3)):
    masks = [eng.alloc_vec(n) for _ in range(C)]
    dmask = eng.alloc_vec(n)
    jobs = [(c, c + 1, 0, n, None, 0, masks[c]) for c in range(C)] + [(C, 0, 0, n, None, 0, dmask)]
    row = []
    for q, parts in ((None, None), (0, None), (1, None), (1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (1, 12)):
        for k in ("FLASHE_CHAIN_QUARTER", "FLASHE_CHAIN_PARTS"):
            os.environ.pop(k, None)
        if q is not None:
            os.environ["FLASHE_CHAIN_QUARTER"] = str(q)
        if parts is not None:
            os.environ["FLASHE_CHAIN_PARTS"] = str(parts)
        row.append(f"{'auto' if q is None else ('half' if q == 0 else 'quarter')}{'' if parts is None else '/' + str(parts)} {timeit(lambda: eng.prf_jobs_dev(0, n, 16, jobs)):.1f}")
    print(f"n={n} C={C}: " + "  ".join(row), flush=True)
    del masks, dmask
