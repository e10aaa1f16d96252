#!/usr/bin/env python3
"""Host-side cost of back-to-back small launches through the C ABI (config-3 shape: 100 combine calls on 61,706-element vectors)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

n, C = 61_706, 100
for b in (128, 23):
    eng = Engine(bytes(range(32)), b)
    L = eng.limbs
    pts = [eng.upload(np.arange(n, dtype=np.uint64)) for _ in range(C)]
    masks = [eng.alloc_vec(n) for _ in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    for with_prf in (False, True):
        for rep in range(4):
            eng.sync()
            t0 = time.perf_counter()
            if with_prf:
                eng.prf_jobs_dev(rep, n, 16, [(c, c + 1, 0, n, None, 0, masks[c]) for c in range(C)])
            t1 = time.perf_counter()
            for c in range(C):
                eng.combine_dev(n, pts[c], 1, masks[c], None, cts[c])
            t2 = time.perf_counter()
            eng.sync()
            t3 = time.perf_counter()
            print(f"b={b} with_prf={with_prf} rep={rep}: prf call {1e3 * (t1 - t0):.3f} ms, 100 combine calls {1e3 * (t2 - t1):.3f} ms, sync {1e3 * (t3 - t2):.3f} ms")
