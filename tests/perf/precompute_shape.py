#!/usr/bin/env python3
"""The mask precompute launch of BASELINE config 3 taken apart: the chain of 100 clients' mask differences alone, the decrypt mask
difference alone, both in one launch (what bench.py --config 3 times), and the pieces-per-chain knob on the combined launch."""
import os
import sys

import numpy as np

os.environ.setdefault("FLASHE_LIB_NAME", "libflashe_hip_tuning.so")      # the knobs below exist only in the -DFLASHE_TUNING build
os.environ["FLASHE_CHAIN_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = Engine(bytes(range(32)), b)
e0, e1 = eng.event(), eng.event()
n, C, J = 61_706, 100, 16
masks = [eng.alloc_vec(n) for _ in range(C)]
dmask = eng.alloc_vec(n)


def timeit(fn, inner=20, reps=6):
    for _ in range(30):
        fn()
    best = 1e9
    for _ in range(reps):
        eng.record(e0)
        for _ in range(inner):
            fn()
        eng.record(e1)
        best = min(best, eng.elapsed_ms(e0, e1) / inner)
    return best * 1e3


chain = [(c, c + 1, 0, n, None, 0, masks[c]) for c in range(C)]
last = [(C, 0, 0, n, None, 0, dmask)]
for name, jobs in (("chain of 100", chain), ("decrypt difference", last), ("both, one launch", chain + last)):
    for k in ("FLASHE_CHAIN_PARTS", "FLASHE_CHAIN_HALF"):
        os.environ.pop(k, None)
    print(f"{name:22s}: {timeit(lambda: eng.prf_jobs_dev(0, n, J, jobs)):7.1f} us", flush=True)
    if len(jobs) > 1:
        for half in (1, 0):
            for parts in (4, 6, 8, 10, 12, 16):
                os.environ["FLASHE_CHAIN_PARTS"], os.environ["FLASHE_CHAIN_HALF"] = str(parts), str(half)
                print(f"      half={half} parts={parts:2d}: {timeit(lambda: eng.prf_jobs_dev(0, n, J, jobs)):7.1f} us", flush=True)
