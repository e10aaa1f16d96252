#!/usr/bin/env python3
"""Config-5 sparse aggregate alone, back to back (for rocprofv3 / FLASHE_SPAN_PROBE phase probes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import _lib  # noqa: E402
from flashe_amd.engine import Engine  # noqa: E402

if os.environ.get("FLASHE_LIB_NAME"):                      # a build kept from an earlier round lacks the newest entry points: bind what it has
    import ctypes
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    for _k in [k_ for k_ in _lib._SIGNATURES if not hasattr(_probe, k_)]:
        del _lib._SIGNATURES[_k]

total, C = 25_557_032, 50
k = total // 100
eng = Engine(bytes(range(32)), 128, device=0)
rng = [np.random.Generator(np.random.PCG64(2000 + c)) for c in range(C)]
locs = [np.sort(r.choice(total, k, replace=False)).astype(np.uint32) for r in rng]
if os.environ.get("ONE_ALLOC"):
    kp = (k + 63) // 64 * 64
    big_l, big_v = eng.alloc(C * kp * 4), eng.alloc(C * kp * 16)
    host = np.zeros(C * kp, dtype=np.uint32)
    for c in range(C):
        host[c * kp:c * kp + k] = locs[c]
    big_l.upload(host)
    d_loc = [big_l.ptr + c * kp * 4 for c in range(C)]
    d_ct = [big_v.ptr + c * kp * 16 for c in range(C)]
else:
    d_loc = [eng.upload(l) for l in locs]
    d_ct = [eng.alloc_vec(k) for _ in range(C)]
d_agg, d_dec = eng.alloc_vec(total), eng.alloc_vec(total)
evs = [eng.event() for _ in range(3)]
for fused in ((1,) if os.environ.get("DECRYPT_ONLY") else (0, 1)):
    for rep in range(3):
        eng.record(evs[0])
        for _ in range(10):
            if fused:
                eng.sparse_decrypt_dev(1, d_loc, [k] * C, total, 16, d_agg, d_dec, sorted_lists=True)
            else:
                eng.sparse_aggregate_dev(total, d_loc, [k] * C, d_ct, [1 << 31] * C, d_agg, sorted_lists=True)
        eng.record(evs[1])
        try:
            eng.sync()
        except Exception as e:
            print("flag", e)
    print("fused decrypt" if fused else "aggregate", "probe", os.environ.get("FLASHE_SPAN_PROBE", "0"), "us per call", eng.elapsed_ms(evs[0], evs[1]) * 100)

if os.environ.get("FLASHE_SPAN_PROBE") == "9":              # tuning build: where workgroup 0 / wave 0 of span_prf_kernel spends its cycles
    import ctypes
    out = (ctypes.c_ulonglong * 24)()
    fn = eng._lib.flashe_tune_span_prf_cycles
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int], ctypes.c_int
    fn(eng._h, out, 1)
    # (round 5, FLASHE_SPAN_OVERLAP: order in the loop = head, rounds, wait + write-out of the previous span, barrier A, atomics, second pass,
    # barrier B; the per-wave figures = loop head -> end of the wave's rounds)
    names = ["search + round 1", "rounds 2..14", "atomics (+ ct store)", "second pass (crowded spans)", "barrier A (after the write-out)", "wait + write-out of the previous span",
             "barrier B (entries in)", "loop head"]
    tot = sum(out[:8])
    spans = int(os.environ.get("SPANS_PER_CALL", 57)) * 30              # 30 calls of ~57 spans per workgroup (config 5 on 256 CUs)
    print("  head -> end of rounds per wave, cycles per span:", [int(v) // spans for v in out[8:24]])
    print("  cycles per span (wave 0):", tot // spans)
    for n, v in zip(names, out):
        print(f"  {n:28s} {v:14d} ticks  {100.0 * v / max(tot, 1):5.1f} %")
