#!/usr/bin/env python3
"""What the page-locked result pool buys the host-pointer twins: per-call and per-round times at n = 1e7, b = 128 with
FLASHE_HOST_POOL_PINNED = 0 / 1 (one subprocess each, alternated twice), plus the price of page-locking itself and the
device-handle round of the drop-in class.  usage: e2e_pinned.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, %r)
from flashe_amd.engine import SCHEME_DOUBLE, Engine
from flashe_amd import _lib
n, C = 10_000_000, 10
eng = Engine(bytes(range(32)), 128, device=0)
pts = [np.random.Generator(np.random.PCG64(c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
lib = _lib.load()
t0 = time.perf_counter(); p = ctypes.c_void_p(); lib.flashe_host_alloc(160 << 20, ctypes.byref(p)); t_alloc = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); lib.flashe_host_free(p); t_free = (time.perf_counter() - t0) * 1e3
def t(f, reps=4):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
first0 = time.perf_counter(); ct = eng.encrypt(0, 0, SCHEME_DOUBLE, 16, pts[0]); first = (time.perf_counter() - first0) * 1e3
enc, ct = t(lambda: eng.encrypt(0, 0, SCHEME_DOUBLE, 16, pts[0]))
cts = [eng.encrypt(0, c, SCHEME_DOUBLE, 16, pts[c]) for c in range(C)]
agg_ms, agg = t(lambda: eng.aggregate_elem(cts))
dec_ms, dec = t(lambda: eng.decrypt(0, [C], [0], 16, agg))
def rnd():
    cts = [eng.encrypt(0, c, SCHEME_DOUBLE, 16, pts[c]) for c in range(C)]
    agg = eng.aggregate_elem(cts)
    return eng.decrypt(0, [C], [0], 16, agg)
round_ms, _ = t(rnd, 3)
print("pinned=%%s host_alloc(160MB) %%.1f ms free %%.1f | first encrypt %%.1f | encrypt %%.2f  aggregate %%.2f  decrypt %%.2f  round %%.1f ms" %% (
    os.environ.get("FLASHE_HOST_POOL_PINNED", "0"), t_alloc, t_free, first, enc, agg_ms, dec_ms, round_ms))
''' % ROOT

for rep in range(2):
    for pinned in ("0", "1"):
        env = dict(os.environ, FLASHE_HOST_POOL_PINNED=pinned, OMP_WAIT_POLICY="passive")
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-600:], flush=True)
