#!/usr/bin/env python3
"""Launch-shape sweep of the chained PRF kernel on short vectors (BASELINE config 3: 100 clients x 61,706 elements, and a single
61,706-element encrypt): pieces per chain, half tiles vs whole tiles, grid size.  HIP-event times, one process, one box."""
import os
import sys

import numpy as np

os.environ.setdefault("FLASHE_LIB_NAME", "libflashe_hip_tuning.so")      # the knobs below exist only in the -DFLASHE_TUNING build
os.environ["FLASHE_CHAIN_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE, Engine  # noqa: E402

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


# usage: sweep_chain.py            the double-mask shapes of config 3
#        sweep_chain.py single     single-mask shapes (config 5: 50 clients x 255,570 compact entries; one chain, independent streams)
SHAPES = [(61_706, 100), (61_706, 1), (61_706, 10), (1_000_000, 10), (250_000, 100)]
SCHEME = SCHEME_DOUBLE
GRIDS = (None, 128, 64)
if len(sys.argv) > 1 and sys.argv[1] == "single":
    SHAPES, SCHEME, GRIDS = [(255_570, 50), (61_706, 100), (1_000_000, 12)], SCHEME_SINGLE, (None,)
for n, C in SHAPES:
    pts = [eng.upload(np.arange(n, dtype=np.uint64)) for _ in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    idx = list(range(C))
    run = lambda: eng.encrypt_batch_dev(0, idx, SCHEME, n, 16, pts, 1, cts)
    setenv()
    blocks = n * (C if SCHEME == SCHEME_SINGLE else C + 1)
    t = timeit(run)
    print(f"n={n} C={C}: default {t:.1f} us = {blocks / t / 1e3:.1f} G blocks/s", flush=True)
    for half in (1, 0):
        for parts in sorted({1, 2, 3, 4, 6, 8, 12, 16} & set(range(1, C + 1))):
            for grid in GRIDS:
                setenv(half=half, parts=parts, grid=grid)
                print(f"   half={half} parts={parts:2d} grid={grid}: {timeit(run):.1f} us", flush=True)
    setenv()
    t = timeit(run)          # again, with the clock where the sweep left it (the first line is measured on a cold chip)
    print(f"   default again: {t:.1f} us = {blocks / t / 1e3:.1f} G blocks/s", flush=True)
