#!/usr/bin/env python3
"""A lone LeNet-sized encrypt / no-dropout decrypt (61,706 elements, b = 128): kernel time back to back (HIP events) with the
latency form (FLASHE_SMALL_LATENCY=1: job-table kernel, 256-thread workgroups, add and minus block of an element as one pair) and
with the chained kernel's half tiles (=0), alternated as subprocesses.  usage: lone_encrypt.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from flashe_amd.engine import SCHEME_DOUBLE, Engine
from oracle import flashe_oracle as orc
eng = Engine(bytes(range(32)), 128)
out = []
for n in (1000, 61_706, 65_536, 70_000, 400_000):
    pt = np.arange(n, dtype=np.uint64) * 977
    d, c, r = eng.upload(pt), eng.alloc_vec(n), eng.alloc_vec(n)
    enc = lambda: eng.encrypt_dev(5, 3, SCHEME_DOUBLE, n, 16, d, 1, c)
    dec = lambda: eng.decrypt_dev(5, [9], [0], n, 16, c, r)
    enc(); dec()
    assert np.array_equal(c.download(np.uint64, 2 * n).reshape(n, 2), orc.encrypt(bytes(range(32)), 5, 3, "double", 16, 128, pt))
    e0, e1 = eng.event(), eng.event()
    res = []
    for fn in (enc, dec):
        best = 1e9
        for _ in range(5):
            eng.record(e0)
            for _ in range(20): fn()
            eng.record(e1)
            best = min(best, eng.elapsed_ms(e0, e1) / 20)
        res.append(best * 1e3)
    out.append("n=%%d: encrypt %%.1f us, decrypt %%.1f us" %% (n, res[0], res[1]))
print(" | ".join(out))
''' % ROOT
for rep in range(2):
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, FLASHE_LIB_NAME=os.environ.get("FLASHE_LIB_NAME", "libflashe_hip_tuning.so"), FLASHE_SMALL_LATENCY=v, OMP_WAIT_POLICY="passive"), capture_output=True, text=True)
        print(f"FLASHE_SMALL_LATENCY={v}: " + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-500:]), flush=True)
