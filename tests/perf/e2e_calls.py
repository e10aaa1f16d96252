#!/usr/bin/env python3
"""Where the PCIe-inclusive round goes: per-call times of the host-pointer twins at n = 1e7, b = 128."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine

n, C = 10_000_000, 10
eng = Engine(bytes(range(32)), 128, device=0)
pts = [np.random.Generator(np.random.PCG64(c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]


def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r


ms, ct = t(lambda: eng.encrypt(0, 0, SCHEME_DOUBLE, 16, pts[0]))
print(f"encrypt (80 MB up, 160 MB down, fresh output array)  {ms:7.2f} ms")
ms, _ = t(lambda: np.zeros((n, 2), dtype=np.uint64))
print(f"np.zeros((n, 2))                                      {ms:7.2f} ms")
z = np.zeros((n, 2), dtype=np.uint64)
ms, _ = t(lambda: z.fill(1), 1)
print(f"first touch of a fresh 160 MB array (fill)            {ms:7.2f} ms")
cts = [eng.encrypt(0, c, SCHEME_DOUBLE, 16, pts[c]) for c in range(C)]
ms, agg = t(lambda: eng.aggregate_elem(cts))
print(f"aggregate_elem (10 x 160 MB up, 160 MB down)          {ms:7.2f} ms")
ms, dec = t(lambda: eng.decrypt(0, [C], [0], 16, agg))
print(f"decrypt (160 MB up, 160 MB down)                      {ms:7.2f} ms")
d = eng.alloc_vec(n)
ms, _ = t(lambda: d.upload(cts[0]))
print(f"upload 160 MB                                         {ms:7.2f} ms")
ms, _ = t(lambda: d.download(np.uint64, 2 * n))
print(f"download 160 MB (fresh array)                         {ms:7.2f} ms")
