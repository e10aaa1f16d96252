#!/usr/bin/env python3
"""The chained encrypt with and without the local partial aggregate (prf_chain_kernel<.., SUM>), and the two round forms built on
them, alternated inside ONE process (the shader clock moves between processes): HIP-event times per launch / round.
usage: ab_partial_agg.py [n] [C]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = Engine(bytes(range(32)), 128)
pts = [eng.upload(np.random.default_rng(c).integers(0, 2 ** 64, n, dtype=np.uint64)) for c in range(C)]
stride = 2 * n
ct_all = eng.alloc(C * stride * 8)
cts = [ct_all.ptr + 8 * c * stride for c in range(C)]
part, res = eng.alloc_vec(n), eng.alloc_vec(n)
idx = list(range(C))


def enc_plain():
    eng.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, 16, pts, 1, cts)


def enc_sum():
    eng.encrypt_batch_sum_dev(0, idx, SCHEME_DOUBLE, n, 16, pts, 1, cts, part)


def round_two_launch():
    enc_plain()
    eng.aggregate_decrypt_range_dev(0, [C], [0], n, 16, 0, n, cts, part, res)


def round_partial():
    enc_sum()
    eng.decrypt_range_dev(0, [C], [0], n, 16, 0, n, part, res)


e0, e1 = eng.event(), eng.event()
for _ in range(30):
    round_two_launch()
results = {}
for rep in range(6):
    for name, fn in (("encrypt", enc_plain), ("encrypt+sum", enc_sum), ("round two-launch", round_two_launch), ("round partial-agg", round_partial)):
        fn()
        eng.record(e0)
        for _ in range(10):
            fn()
        eng.record(e1)
        results.setdefault(name, []).append(eng.elapsed_ms(e0, e1) / 10)
for name, v in results.items():
    print(f"{name:20s} best {min(v):.4f} ms  median {sorted(v)[len(v) // 2]:.4f} ms   all {[round(x, 4) for x in v]}")
