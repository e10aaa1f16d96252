#!/usr/bin/env python3
"""VERDICT r3 #5, measured: would the reduce fused with the decrypt of its result (int_bits > 64: ten 16-byte operands + two AES blocks
per element) gain from TWO workgroups per CU on half-size (64-KiB) tables, "so that one workgroup streams its operands while the other
runs rounds"?  Tuning build only (libflashe_hip_tuning.so: flashe_tune_reduce_decrypt_probe).  The same simplified loop in three shapes,
alternated in one process, HIP events:
   0  1024 threads, full 128-KiB tables, one workgroup per CU    -- checked against the oracle
   1  2 x 512 threads per CU on 64-KiB tables (tables 2 / 3 alias 0 / 1: a TIMING PROBE, results wrong; a real two-table AES adds a
      rotate per aliased lookup on top, so this is the upper bound of what the split can buy)
   2  1024 threads on the 64-KiB tables (separates "two workgroups" from "smaller tables")
and, for reference, the product's fused launch (flashe_aggregate_decrypt_range_dev)."""
import ctypes
import os
import sys

os.environ.setdefault("FLASHE_LIB_NAME", "libflashe_hip_tuning.so")
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import _lib  # noqa: E402
from flashe_amd.engine import Engine  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes(range(32))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lib = _lib.load()
fn = lib.flashe_tune_reduce_decrypt_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
               ctypes.c_uint64, ctypes.c_void_p]
eng = Engine(KEY, 128, device=0)
orc.build()
for C in (10, 4):
    rng = np.random.Generator(np.random.PCG64(C))
    cts = [rng.integers(0, 2 ** 64, (n, 2), dtype=np.uint64) for _ in range(C)]
    # equally spaced in ONE allocation: the shape the product's one-launch form takes (ShardedRound lays its ciphertexts out like this)
    stride = (2 * n + 1) // 2 * 2
    big = eng.alloc(C * stride * 8)
    for c in range(C):
        big.upload_at(c * stride * 8, cts[c])
    dev = [big.ptr + c * stride * 8 for c in range(C)]
    ptrs = (ctypes.c_void_p * C)(*dev)
    out = eng.alloc_vec(n)
    it = 5

    def probe(variant):
        eng._check(fn(eng._h, variant, it, C, 0, C, ptrs, n, out.ptr))

    def product():
        eng.aggregate_decrypt_range_dev(it, [C], [0], n, 16, 0, n, dev, None, out)

    probe(0)
    want = orc.decrypt(KEY, it, [C], [0], 16, 128, orc.aggregate_elem(cts, 128))
    assert np.array_equal(out.download(np.uint64, 2 * n).reshape(n, 2), want), "variant 0 must be correct"
    runs = {"product launch": product, "probe 0: 1024 thr, 128 KiB": lambda: probe(0), "probe 1: 2 x 512 thr, 64 KiB (timing only)": lambda: probe(1),
            "probe 2: 1024 thr, 64 KiB (timing only)": lambda: probe(2)}
    best = {k: 1e9 for k in runs}
    e0, e1 = eng.event(), eng.event()
    for rep in range(12):
        for name, f in runs.items():
            f()                                   # (one untimed launch of this shape first: the clock and the caches see the same history)
            eng.record(e0)
            f()
            eng.record(e1)
            eng.sync()
            best[name] = min(best[name], eng.elapsed_ms(e0, e1))
    for name, v in best.items():
        print(f"C = {C:2d}, n = {n}: {name:48s} {v:7.4f} ms", flush=True)
