#!/usr/bin/env python3
"""Back-to-back launch times (HIP events) of the PRF launches on short vectors: BASELINE config 3 shapes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

for b in (128, 23):
    eng = Engine(bytes(range(32)), b)
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

    for n, C in [(61_706, 1), (61_706, 10), (61_706, 100), (1_000_000, 1)]:
        pts = [eng.upload(np.arange(n, dtype=np.uint64)) for _ in range(C)]
        cts = [eng.alloc_vec(n) for _ in range(C)]
        agg, dec = eng.alloc_vec(n), eng.alloc_vec(n)
        t_enc = timeit(lambda: eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, pts, 1, cts))
        t_dec = timeit(lambda: eng.decrypt_dev(0, [C], [0], n, 16, cts[0], dec))
        t_agg = timeit(lambda: eng.aggregate_elem_dev(cts, n, agg))
        print(f"b={b} n={n} C={C}: encrypt batch {t_enc:.1f} us, decrypt {t_dec:.1f} us, aggregate {t_agg:.1f} us", flush=True)
