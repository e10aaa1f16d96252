#!/usr/bin/env python3
"""Does the chained encrypt's AES rate depend on the bytes it moves per block?  b = 128, ten 1e7-element vectors, plaintext given as
8-byte words (24 B per block) and as 16-byte containers (32 B per block, what a b = 64 launch moves per block); and the b = 64 launch
itself.  HIP-event times, alternated in one process."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

n, C = 10_000_000, 10
e128, e64 = Engine(bytes(range(32)), 128), Engine(bytes(range(32)), 64)
rng = np.random.default_rng(0)
p1 = [e128.upload(rng.integers(0, 2 ** 64, n, dtype=np.uint64)) for _ in range(C)]
p2 = [e128.upload(rng.integers(0, 2 ** 64, (n, 2), dtype=np.uint64)) for _ in range(C)]
c128 = [e128.alloc_vec(n) for _ in range(C)]
c64 = [e64.alloc_vec(n) for _ in range(C)]
idx = list(range(C))
cases = {"b=128, 8-byte plaintext (24 B/block)": (e128, lambda: e128.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, 16, p1, 1, c128), 11 * n),
         "b=128, 16-byte plaintext (32 B/block)": (e128, lambda: e128.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, 16, p2, 2, c128), 11 * n),
         "b=64 (32 B/block)": (e64, lambda: e64.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, 16, p1, 1, c64), 11 * n // 2)}
res = {k: [] for k in cases}
for k, (eng, fn, _) in cases.items():
    for _ in range(5):
        fn()
for rep in range(5):
    for k, (eng, fn, blocks) in cases.items():
        e0, e1 = eng.event(), eng.event()
        fn()
        eng.record(e0)
        for _ in range(10):
            fn()
        eng.record(e1)
        res[k].append(eng.elapsed_ms(e0, e1) / 10)
for k, (_, _, blocks) in cases.items():
    best = min(res[k])
    print(f"{k:42s} best {best:.4f} ms = {blocks / best / 1e6:.1f} G blocks/s   all {[round(v, 4) for v in res[k]]}")
