#!/usr/bin/env python3
"""Launch-shape sweep of the chained PRF kernel on short vectors (BASELINE config 3: 100 clients x 61,706 elements, and a single
61,706-element encrypt): pieces per chain, half tiles vs whole tiles, grid size.  HIP-event times, one process, one box."""
import os
import sys

import numpy as np

os.environ["FLASHE_CHAIN_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

eng = Engine(bytes(range(32)), 128)
e0, e1 = eng.event(), eng.event()


def timeit(fn, inner=20, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        eng.record(e0)
        for _ in range(inner):
            fn()
        eng.record(e1)
        best = min(best, eng.elapsed_ms(e0, e1) / inner)
    return best * 1e3


def setenv(**kw):
    for k in ("FLASHE_CHAIN_HALF", "FLASHE_CHAIN_PARTS", "FLASHE_CHAIN_GRID"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        if v is not None:
            os.environ["FLASHE_CHAIN_" + k.upper()] = str(v)


for n, C in [(61_706, 100), (61_706, 1), (61_706, 10), (1_000_000, 10), (250_000, 100)]:
    pts = [eng.upload(np.arange(n, dtype=np.uint64)) for _ in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    idx = list(range(C))
    run = lambda: eng.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, 16, pts, 1, cts)
    setenv()
    print(f"n={n} C={C}: default {timeit(run):.1f} us", flush=True)
    for half in (1, 0):
        for parts in sorted({1, 2, 4, 8, 12, 16} & set(range(1, C + 1))):
            for grid in (None, 128, 64):
                setenv(half=half, parts=parts, grid=grid)
                print(f"   half={half} parts={parts:2d} grid={grid}: {timeit(run):.1f} us", flush=True)
