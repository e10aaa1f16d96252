#!/usr/bin/env python3
"""Top-k sparsifier (flashe_sparsify_dev, Client.sparsify jzf_aggregator.py:578-623 for one layer) on ResNet-50-sized input:
25,557,032 values, k = 1 %, float32 and float64, with the residual; HIP-event time per call and bytes per second against the
stages' algorithmic traffic ((digits + 2) reads of x, one read + write of the residual, the k outputs)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

eng = Engine(bytes(32), 64)
e0, e1 = eng.event(), eng.event()
for n in (25_557_032, 2_359_296, 61_706):
    k = max(1, n // 100)
    for dt in (np.float32, np.float64):
        x = (np.random.default_rng(1).standard_normal(n) * 0.05).astype(dt)
        dx, dres = eng.upload(x), eng.upload(np.zeros(n, dtype=dt))
        dloc, dval = eng.alloc(4 * k + 16), eng.alloc(x.itemsize * k + 16)
        run = lambda: eng.sparsify_dev(n, k, dx, dt == np.float64, dres, dloc, dval)
        for _ in range(5):
            run()
        best = 1e9
        for rep in range(5):
            eng.record(e0)
            for _ in range(10):
                run()
            eng.record(e1)
            best = min(best, eng.elapsed_ms(e0, e1) / 10)
        digits = x.itemsize
        alg = (digits + 2) * n * x.itemsize + 2 * n * x.itemsize + k * (4 + x.itemsize)
        print(f"n={n} k={k} {np.dtype(dt).name}: {best * 1e3:8.1f} us  = {alg / best / 1e6:7.1f} GB/s of its own {alg / 1e6:.0f} MB", flush=True)
