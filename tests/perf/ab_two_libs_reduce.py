#!/usr/bin/env python3
"""Two BUILDS of the library alternated inside one process: the reduce fused with the decrypt of its result (ten 1e7-element
128-bit ciphertexts in, one decrypted vector out) and the whole two-launch round, HIP-event times.
usage: ab_two_libs_reduce.py <other .so in flashe_amd/> [more .so ...] [C ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import _lib  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402


def engine_from(name, b):
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "flashe_amd", name)
    return Engine(bytes(range(32)), b)


others = [a for a in sys.argv[1:] if a.endswith(".so")]
n, b = 10_000_000, 128
for C in [int(v) for v in sys.argv[1:] if not v.endswith(".so")] or [10, 4, 16]:
    engs = {"libflashe_hip.so": engine_from("libflashe_hip.so", b)}
    for o in others:
        engs[o] = engine_from(o, b)
    runs = {}
    for name, eng in engs.items():
        pts = [eng.upload(np.random.default_rng(c).integers(0, 2 ** 62, n, dtype=np.uint64)) for c in range(C)]
        cts = eng.alloc_vec(n * C)
        views = [cts.ptr + 16 * n * c for c in range(C)]
        out = eng.alloc_vec(n)
        enc = (lambda e=eng, p=pts, v=views: e.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, p, 1, v))
        red = (lambda e=eng, v=views, o=out: e.aggregate_decrypt_range_dev(0, [C], [0], n, 16, 0, n, v, None, o))
        enc()
        psum = eng.alloc_vec(n)
        part = (lambda e=eng, p=pts, v=views, q=psum, o=out: (e.encrypt_batch_sum_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, p, 1, v, q),
                                                               e.decrypt_dev(0, [C], [0], n, 16, q, o)))
        runs[name] = (eng, red, (lambda a=enc, r=red: (a(), r())), eng.event(), eng.event(), (pts, cts, out, psum), part)
    for what, idx in (("reduce+decrypt", 1), ("round", 2), ("partial-agg round", 6)):
        res = {k: [] for k in engs}
        for rep in range(6):
            for name, r in runs.items():
                eng, run, e0, e1 = r[0], r[idx], r[3], r[4]
                run(); run()
                eng.record(e0)
                for _ in range(10):
                    run()
                eng.record(e1)
                res[name].append(eng.elapsed_ms(e0, e1) / 10)
        print(f"C={C} {what}: " + " | ".join(f"{k} {min(v):.4f} ms (all {[round(x, 3) for x in v]})" for k, v in res.items()), flush=True)
