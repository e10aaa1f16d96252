#!/usr/bin/env python3
"""int_bits <= 64: batched encrypt launches with (FLASHE_CHAIN=1) and without (=0) stream sharing between consecutive clients,
HIP-event times, interleaved in one gpurun call.  usage: ab_small_chain.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from flashe_amd.engine import Engine, SCHEME_DOUBLE
from oracle import flashe_oracle as orc
out = []
for b, n, C in [(64, 10_000_000, 10), (48, 10_000_000, 10), (40, 10_000_000, 10), (32, 10_000_000, 10), (25, 10_000_000, 10), (20, 10_000_000, 10), (16, 10_000_000, 10), (23, 61_706, 100), (64, 10_000_000, 1)]:
    eng = Engine(bytes(range(32)), b)
    host = [np.random.default_rng(c).integers(0, 2**min(b, 63), n, dtype=np.uint64) for c in range(min(C, 3))]
    pts = [eng.upload(host[c %% len(host)]) for c in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    run = lambda: eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, pts, 1, cts)
    for _ in range(3): run()
    k = min(n, 200000)
    got = cts[C - 1].download(np.uint64, n)[:k]
    want = orc.encrypt(bytes(range(32)), 0, C - 1, "double", 16, b, host[(C - 1) %% len(host)])[:k, 0]
    if "NOCHECK" not in __import__("os").environ: assert np.array_equal(got, want), "WRONG RESULT"
    e0, e1 = eng.event(), eng.event()
    best = 1e9
    for rep in range(5):
        eng.record(e0)
        for _ in range(5): run()
        eng.record(e1)
        best = min(best, eng.elapsed_ms(e0, e1) / 5)
    m = 128 // b
    blocks = 2 * C * ((n + m - 1) // m)
    out.append("b=%%d n=%%d C=%%d: %%.4f ms (%%.1f G blocks/s as 2C streams)" %% (b, n, C, best, blocks / best / 1e6))
print(" | ".join(out))
''' % ROOT

VAR = sys.argv[1] if len(sys.argv) > 1 else "FLASHE_CHAIN"        # or FLASHE_SMALL_DIRECT
for rnd in range(2):
    for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "0")):
        env = dict(os.environ, OMP_WAIT_POLICY="passive", FLASHE_LIB_NAME=os.environ.get("FLASHE_LIB_NAME", "libflashe_hip_tuning.so"))
        env[VAR] = v
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"round {rnd} {VAR}={v}: " + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-400:]), flush=True)
