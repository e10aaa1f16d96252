#!/usr/bin/env python3
"""The headline launch -- ten 1e7-element encrypts as one chain + their partial aggregate (prf_chain_kernel<1024, SUM>), int_bits 128 --
and the decrypt of that vector under several BUILDS of the library alternated inside one process; outputs compared byte for byte.
usage: ab_chain_libs.py <.so in flashe_amd/> <.so> [...]     (AB_REPS alternations, default 6; AB_BITS, default 128; AB_N / AB_C: vector
       length and clients, e.g. 61706 / 100 = config 3's shape)"""
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


libs = sys.argv[1:]
n, C, J, K = int(os.environ.get("AB_N", "10000000")), int(os.environ.get("AB_C", "10")), 16, 10
b = int(os.environ.get("AB_BITS", "128"))
reps = int(os.environ.get("AB_REPS", "6"))
st = {}
for name in libs:
    eng = engine_from(name, b)
    pts = [eng.upload(np.random.default_rng(c).integers(0, 2 ** 63, n, dtype=np.uint64)) for c in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    dsum, dec = eng.alloc_vec(n), eng.alloc_vec(n)
    st[name] = (eng, pts, cts, dsum, dec, [eng.event() for _ in range(2 * K + 1)])
idx = list(range(C))
res = {name: [] for name in libs}
for rep in range(reps + 1):
    for name in libs:
        eng, pts, cts, dsum, dec, ev = st[name]
        for it in range(3):
            eng.encrypt_batch_sum_dev(it, idx, SCHEME_DOUBLE, n, J, pts, 1, cts, dsum)
            eng.decrypt_dev(it, [C], [0], n, J, dsum, dec)
        eng.record(ev[0])
        for k in range(K):
            eng.encrypt_batch_sum_dev(k, idx, SCHEME_DOUBLE, n, J, pts, 1, cts, dsum)
            eng.record(ev[2 * k + 1])
            eng.decrypt_dev(k, [C], [0], n, J, dsum, dec)
            eng.record(ev[2 * k + 2])
        eng.sync()
        if rep:                                              # (the first alternation warms the clocks)
            res[name].append((np.mean([eng.elapsed_ms(ev[2 * k], ev[2 * k + 1]) for k in range(K)]), np.mean([eng.elapsed_ms(ev[2 * k + 1], ev[2 * k + 2]) for k in range(K)])))
ref = None
for name in libs:
    eng, pts, cts, dsum, dec, ev = st[name]
    got = (cts[3].download(np.uint64, 2 * n if b > 64 else n).tobytes(), dec.download(np.uint64, 2 * n if b > 64 else n).tobytes())
    ref = ref or got
    a = np.array(res[name])
    print(f"{name:28s} chain + sum {a[:, 0].min():.4f} (med {np.median(a[:, 0]):.4f})   decrypt {a[:, 1].min():.4f} (med {np.median(a[:, 1]):.4f})   "
          f"round {np.median(a[:, 0]) + np.median(a[:, 1]):.4f} ms   identical to the first build: {got == ref}", flush=True)
