#!/usr/bin/env python3
"""Encrypt rate of the b <= 64 path (m = 128 // b elements per AES block) at n = 1e7: single vector and ten batched."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

n, C, J = 10_000_000, 10, 16
rng = np.random.default_rng(1)
for b in (64, 32, 23, 20, 8):
    eng = Engine(bytes(range(32)), b)
    pts = [eng.upload(rng.integers(0, 2 ** min(b, 63), n, dtype=np.uint64)) for _ in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    e0, e1 = eng.event(), eng.event()

    def t(fn, inner):
        best = 1e9
        for rep in range(4):
            eng.record(e0)
            for _ in range(inner):
                fn()
            eng.record(e1)
            best = min(best, eng.elapsed_ms(e0, e1) / inner)
        return best
    one = t(lambda: eng.encrypt_dev(0, 3, SCHEME_DOUBLE, n, J, pts[3], 1, cts[3]), 10)
    ten = t(lambda: eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, J, pts, 1, cts), 3)
    m = 128 // b
    blocks = 2 * ((n + m - 1) // m)
    print(f"b={b} m={m}: one vector {one:.4f} ms ({blocks / one / 1e6:.1f} G blocks/s, {16 * n / one / 1e9:.2f} TB/s); "
          f"ten batched {ten:.4f} ms ({C * blocks / ten / 1e6:.1f} G blocks/s, {C * 16 * n / ten / 1e9:.2f} TB/s)", flush=True)
