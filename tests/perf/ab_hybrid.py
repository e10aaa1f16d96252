#!/usr/bin/env python3
"""Single-vector encrypt (n = 1e7, b = 128, double mask): table backend vs the hybrid backend (a share of the range on
the bit-sliced kernel, co-scheduled on a second stream) for several shares."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from flashe_amd.engine import Engine, SCHEME_DOUBLE
from oracle import flashe_oracle as orc
backend = int(sys.argv[1])
eng = Engine(bytes(range(32)), 128)
eng.set_prf_backend(backend)
n = 10_000_000
host = np.random.default_rng(3).integers(0, 2**64, n, dtype=np.uint64)
pt, ct = eng.upload(host), eng.alloc_vec(n)
for _ in range(3): eng.encrypt_dev(0, 3, SCHEME_DOUBLE, n, 16, pt, 1, ct)
got = ct.download(np.uint64, 2 * n).reshape(n, 2)
want = orc.encrypt(bytes(range(32)), 0, 3, "double", 16, 128, host)
assert np.array_equal(got, want), "WRONG RESULT"
e0, e1 = eng.event(), eng.event()
best = 1e9
for rep in range(5):
    eng.record(e0)
    for _ in range(10): eng.encrypt_dev(0, 3, SCHEME_DOUBLE, n, 16, pt, 1, ct)
    eng.record(e1)
    best = min(best, eng.elapsed_ms(e0, e1) / 10)
print("%%.4f" %% best)
''' % ROOT

for rnd in range(2):
    for backend, pm in ((1, 0), (3, 50), (3, 100), (3, 150), (3, 200)):
        env = dict(os.environ, FLASHE_LIB_NAME=os.environ.get("FLASHE_LIB_NAME", "libflashe_hip_bitslice.so"), FLASHE_HYBRID_BS_PERMILLE=str(pm))
        out = subprocess.run([sys.executable, "-c", CODE, str(backend)], env=env, capture_output=True, text=True)
        res = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-300:]
        print(f"round {rnd} backend={'table' if backend == 1 else 'hybrid'} bitsliced share {pm / 10:.0f} %: {res} ms", flush=True)
