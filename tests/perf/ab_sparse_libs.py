#!/usr/bin/env python3
"""Config 5's two fused sparse passes under two BUILDS of the library alternated inside one process (same box, same clock):
flashe_sparse_encrypt_aggregate_dev (50 encrypts + the aggregate of their uploads) and flashe_sparse_decrypt_dev, HIP-event times, and
the results of the two builds compared byte for byte (the GPU suite compares the product build with the oracle).
usage: ab_sparse_libs.py <.so in flashe_amd/> [more .so ...]      e.g. after  make -C flashe_amd/csrc ab ABFLAGS=-DFLASHE_SPAN_OVERLAP=0
       (the first library named is the reference of the byte comparison; AB_REPS = alternations, default 6; `make ab` builds carry
       -DFLASHE_TUNING, so compare them with libflashe_hip_tuning.so rather than with the product library)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import _lib  # noqa: E402
from flashe_amd.engine import Engine  # noqa: E402


_ALL_SIGNATURES = dict(_lib._SIGNATURES)


def engine_from(name, b):
    import ctypes
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "flashe_amd", name)
    # (a build kept from an earlier round lacks the entry points added since: bind what it has -- this tool only, the package itself
    # insists on every symbol of the header)
    probe = ctypes.CDLL(_lib.LIB_PATH)
    _lib._SIGNATURES.clear()
    _lib._SIGNATURES.update({k: v for k, v in _ALL_SIGNATURES.items() if hasattr(probe, k)})
    return Engine(bytes(range(32)), b)


libs = sys.argv[1:] if len(sys.argv) > 2 else ["libflashe_hip.so", sys.argv[1]]
reps = int(os.environ.get("AB_REPS", 6))
total, C, b, J = int(os.environ.get("AB_TOTAL", 25_557_032)), int(os.environ.get("AB_CLIENTS", 50)), 128, 16
k = total // 100
rng = [np.random.Generator(np.random.PCG64(2000 + c)) for c in range(C)]
locs = [np.sort(r.choice(total, k, replace=False)).astype(np.uint32) for r in rng]
vals = [r.integers(0, 2 ** 64, k, dtype=np.uint64) for r in rng]
zero = 1 << 31
runs, outs = {}, {}
for name in libs:
    eng = engine_from(name, b)
    d_loc, d_val = [eng.upload(l) for l in locs], [eng.upload(v) for v in vals]
    d_ct = [eng.alloc_vec(k) for _ in range(C)]
    d_agg, d_dec = eng.alloc_vec(total), eng.alloc_vec(total)
    t_loc, t_val, t_ct, t_k = eng.ptr_table(d_loc), eng.ptr_table(d_val), eng.ptr_table(d_ct), eng.u64_table([k] * C)
    t_zero, idx = eng.zeros_table([zero] * C), list(range(C))
    bounds = eng.span_bounds(total, t_loc, t_k)

    def enc(it, eng=eng, t_loc=t_loc, t_k=t_k, t_val=t_val, t_zero=t_zero, t_ct=t_ct, d_agg=d_agg, bounds=bounds):
        eng.sparse_encrypt_aggregate_dev(it, idx, t_loc, t_k, t_val, 1, t_zero, total, J, t_ct, d_agg, bounds=bounds)

    def dec(it, eng=eng, t_loc=t_loc, t_k=t_k, d_agg=d_agg, d_dec=d_dec, bounds=bounds):
        eng.sparse_decrypt_dev(it, t_loc, t_k, total, J, d_agg, d_dec, sorted_lists=True, bounds=bounds)

    def rebound(eng=eng, t_loc=t_loc, t_k=t_k, bounds=bounds):
        bounds.recompute(t_loc, t_k)

    runs[name] = (eng, enc, dec, rebound, [eng.event() for _ in range(4)], (d_ct, d_agg, d_dec))
    enc(0); dec(0); eng.sync()
    outs[name] = (d_agg.download(np.uint64, 2 * total), d_dec.download(np.uint64, 2 * total), d_ct[0].download(np.uint64, 2 * k), d_ct[C - 1].download(np.uint64, 2 * k))
a = outs[libs[0]]
same = all(np.array_equal(x, y) for o in outs.values() for x, y in zip(a, o))
# the decrypted dense vector is the plain sum: low limb check (the GPU suite and bench.py check every limb against the oracle)
want = np.full(total, np.uint64((C * zero) & (2 ** 64 - 1)), dtype=np.uint64)
for c in range(C):
    want[locs[c]] += vals[c] - np.uint64(zero)
print("results identical between the builds:", same, "| round trip (low limb) ok:", bool(np.array_equal(a[1][0::2], want)), flush=True)
res = {n: {"bounds": [], "enc": [], "dec": []} for n in runs}
for rep in range(reps):
    for name, (eng, enc, dec, rebound, ev, _bufs) in runs.items():
        for it in range(3):
            rebound(); enc(it); dec(it)
        N = 10
        eng.record(ev[0])
        for it in range(N):
            rebound()
        eng.record(ev[1])
        for it in range(N):
            enc(it)
        eng.record(ev[2])
        for it in range(N):
            dec(it)
        eng.record(ev[3])
        eng.sync()
        for key, i in (("bounds", 0), ("enc", 1), ("dec", 2)):
            res[name][key].append(eng.elapsed_ms(ev[i], ev[i + 1]) / N)
for name, r in res.items():
    print(f"{name:28s} " + "  ".join(f"{k_} {min(v):.4f} (med {sorted(v)[len(v) // 2]:.4f})" for k_, v in r.items())
          + f"  round {min(r['bounds']) + min(r['enc']) + min(r['dec']):.4f} ms", flush=True)
