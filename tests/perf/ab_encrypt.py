#!/usr/bin/env python3
"""A/B timing of the batched encrypt launch (ten 1e7-element vectors, b = 128, double mask) under different
environment-selected kernel variants, interleaved inside ONE process group on ONE GPU box (boxes of the pool differ
by several percent, so only such interleaved comparisons are meaningful).
usage: ab_encrypt.py VAR=val[,val...]    e.g.  ab_encrypt.py FLASHE_TT_TABLES=4,2"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from flashe_amd.engine import Engine, SCHEME_DOUBLE
from oracle import flashe_oracle as orc
eng = Engine(bytes(range(32)), 128)
n, C = 10_000_000, 10
host = [np.random.default_rng(c).integers(0, 2**64, n, dtype=np.uint64) for c in range(C)]
pts = [eng.upload(h) for h in host]
cts = [eng.alloc_vec(n) for _ in range(C)]
for _ in range(3): eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, pts, 1, cts)
got = cts[7].download(np.uint64, 400000).reshape(200000, 2)
assert np.array_equal(got, orc.encrypt(bytes(range(32)), 0, 7, "double", 16, 128, host[7][:200000])), "WRONG RESULT"
e0, e1 = eng.event(), eng.event()
best = 1e9
for rep in range(5):
    eng.record(e0)
    for _ in range(5): eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, pts, 1, cts)
    eng.record(e1)
    best = min(best, eng.elapsed_ms(e0, e1) / 5)
def timeit(fn, inner):
    b = 1e9
    for rep in range(5):
        eng.record(e0)
        for _ in range(inner): fn()
        eng.record(e1)
        b = min(b, eng.elapsed_ms(e0, e1) / inner)
    return b
one = timeit(lambda: eng.encrypt_dev(0, 3, SCHEME_DOUBLE, n, 16, pts[3], 1, cts[3]), 10)
q = 2_500_032
rng = timeit(lambda: eng.encrypt_range_dev(0, 3, SCHEME_DOUBLE, n, 16, q, q, pts[3].ptr + 8 * q, 1, cts[3].ptr + 16 * q), 20)
print("%%.4f  one vector %%.4f  range of 2.5M %%.4f" %% (best, one, rng))
''' % ROOT


def main():
    var, vals = sys.argv[1].split("=")
    vals = vals.split(",")
    for rnd in range(3):
        for v in vals:
            env = dict(os.environ)
            env[var] = v
            out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
            res = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-300:]
            print(f"round {rnd}  {var}={v}: {res} (ms per 10-vector launch, per single-vector launch, per 2.5M-element range launch)")


if __name__ == "__main__":
    main()
