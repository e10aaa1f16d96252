#!/usr/bin/env python3
"""The compact-layout round (int_bits <= 32, uint32 vectors: ten 1e7-element clients, chained encrypt + reduce fused with the decrypt)
under several BUILDS of the library alternated inside one process; the builds' ciphertexts and results are compared byte for byte and
client 3's ciphertext with the oracle.
usage: ab_compact_libs.py <.so in flashe_amd/> <.so> [...]     bits from AB_BITS (default "20 23 16"), AB_LAYOUT=u64 for the one-limb layout,
       AB_SUM=1: the encrypt launch also writes the sum of its ciphertexts and the second launch decrypts that one vector"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import _lib  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402


def engine_from(name, b):
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "flashe_amd", name)
    return Engine(bytes(range(32)), b)


libs = sys.argv[1:]
n, C, J = int(os.environ.get("AB_N", 10_000_000)), 10, 16
u64 = os.environ.get("AB_LAYOUT") == "u64"
orc.build()
for b in [int(v) for v in os.environ.get("AB_BITS", "20 23 16").split()]:
    host = [np.random.default_rng(c).integers(0, 2 ** min(b, 16), n, dtype=np.uint64) for c in range(C)]
    runs, outs = {}, {}
    for name in libs:
        eng = engine_from(name, b)
        idx = list(range(C))
        if u64:
            pt = [eng.upload(h) for h in host]
            ct = [eng.alloc_vec(n) for _ in range(C)]
            out = eng.alloc_vec(n)
            enc = lambda eng=eng, pt=pt, ct=ct: eng.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, J, pt, 1, ct)             # noqa: E731
            dec = lambda eng=eng, ct=ct, out=out: eng.aggregate_decrypt_range_dev(0, [C], [0], n, J, 0, n, ct, None, out)    # noqa: E731
            dt = np.uint64
        else:
            pt = [eng.upload(h.astype(np.uint32)) for h in host]
            ct = [eng.alloc(4 * n) for _ in range(C)]
            out = eng.alloc(4 * n)
            enc = lambda eng=eng, pt=pt, ct=ct: eng.encrypt_batch_u32_dev(0, idx, SCHEME_DOUBLE, n, J, pt, ct)              # noqa: E731
            dec = lambda eng=eng, ct=ct, out=out: eng.aggregate_decrypt_u32_dev(0, [C], [0], n, J, 0, n, ct, None, out, 4)  # noqa: E731
            if os.environ.get("AB_SUM"):          # what bench.py --layout u32 times: the encrypts + their sum, then the decrypt of that vector
                ds = eng.alloc(4 * n)
                enc = lambda eng=eng, pt=pt, ct=ct, ds=ds: eng.encrypt_batch_sum_u32_dev(0, idx, SCHEME_DOUBLE, n, J, pt, ct, ds)          # noqa: E731
                dec = lambda eng=eng, ds=ds, out=out: eng.aggregate_decrypt_u32_dev(0, [C], [0], n, J, 0, n, [ds], None, out, 4)           # noqa: E731
            dt = np.uint32
        enc(); dec(); eng.sync()
        outs[name] = (ct[3].download(dt, n), ct[C - 1].download(dt, n), out.download(dt, n))
        runs[name] = (eng, enc, dec, [eng.event() for _ in range(3)])
    ref = outs[libs[0]]
    same = all(np.array_equal(x, y) for o in outs.values() for x, y in zip(ref, o))
    want = orc.encrypt(bytes(range(32)), 0, 3, "double", J, b, host[3])[:, 0]
    ok = np.array_equal(ref[0].astype(np.uint64), want) and np.array_equal(ref[2].astype(np.uint64), sum(host) & np.uint64((1 << b) - 1))
    res = {k: [] for k in libs}
    for rep in range(int(os.environ.get("AB_REPS", 6))):
        for name, (eng, enc, dec, e) in runs.items():
            for _ in range(3):
                enc(); dec()
            N = 10
            eng.record(e[0])
            for _ in range(N):
                enc()
            eng.record(e[1])
            for _ in range(N):
                dec()
            eng.record(e[2])
            eng.sync()
            res[name].append((eng.elapsed_ms(e[0], e[1]) / N, eng.elapsed_ms(e[1], e[2]) / N))
    print(f"b={b} ({'uint64' if u64 else 'uint32'} vectors): builds identical {same}, oracle / round trip ok {ok}", flush=True)
    for name, v in res.items():
        te, td = min(x[0] for x in v), min(x[1] for x in v)
        print(f"   {name:28s} encrypt x{C} {te:.4f} + reduce/decrypt {td:.4f} = {te + td:.4f} ms", flush=True)
