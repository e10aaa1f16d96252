#!/usr/bin/env python3
"""Object array -> limbs (flashe_amd/csrc/pyconv.c) with one and with all host threads, and limbs -> object array (serial: it
allocates Python ints), 4.2 M elements."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import numpy as np, time, sys, os
sys.path.insert(0, %r)
from flashe_amd import cipher as cm
n = 4_194_304
vals = np.random.default_rng(0).integers(0, 2 ** 20, n, dtype=np.uint64)
obj = np.array([int(v) + (1 << 40) for v in vals], dtype=object)          # distinct int objects, as a ciphertext has
t_in, t_out = [], []
for rep in range(6):
    t0 = time.perf_counter(); a, k = cm._to_limbs(obj, 1); t1 = time.perf_counter(); o = cm._from_limbs(a, k); t2 = time.perf_counter()
    t_in.append((t1 - t0) * 1e3); t_out.append((t2 - t1) * 1e3)
assert int(o[5]) == int(obj[5])
print("FLASHE_PYCONV_THREADS=%%s: ints -> limbs %%.1f ms, limbs -> ints %%.1f ms (best of 6)" %% (os.environ.get("FLASHE_PYCONV_THREADS", "default (usable CPUs, at most 16)"), min(t_in), min(t_out)))
''' % ROOT
for t in ("1", "4", "16", "1", "16", None):
    env = dict(os.environ)
    env.pop("FLASHE_PYCONV_THREADS", None)
    if t:
        env["FLASHE_PYCONV_THREADS"] = t
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or "ERR " + r.stderr[-400:], flush=True)
