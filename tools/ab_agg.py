#!/usr/bin/env python3
"""Streaming rate of the element-wise reduce for several operand counts (one process, one box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flashe_amd.engine import Engine  # noqa: E402

eng = Engine(bytes(range(32)), 128)
rng = np.random.default_rng(1)
res = []
for n, C in ((10_000_000, 2), (10_000_000, 4), (10_000_000, 10), (10_000_000, 24), (4_000_000, 50)):
    base = eng.upload(rng.integers(0, 2 ** 64, 2 * n, dtype=np.uint64))
    src = [base] + [eng.upload(rng.integers(0, 2 ** 64, 2 * n, dtype=np.uint64)) for _ in range(min(C, 12) - 1)]
    src = (src * 6)[:C]                      # operands may repeat: the kernel streams them all the same
    out = eng.alloc_vec(n)
    e0, e1 = eng.event(), eng.event()
    best = 1e9
    for rep in range(4):
        eng.record(e0)
        for _ in range(10):
            eng.aggregate_elem_dev(src, n, out)
        eng.record(e1)
        best = min(best, eng.elapsed_ms(e0, e1) / 10)
    res.append("C=%d: %.3f ms %.2f TB/s" % (C, best, 16 * (C + 1) * n / best / 1e9))
    del src, base, out
print("; ".join(res))
