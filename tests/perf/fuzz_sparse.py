#!/usr/bin/env python3
"""One-off differential fuzz of the sparse kernels (span bounds / span reduce, fused sparse decrypt, scatter forms) against the
oracle: random totals around span boundaries, client counts up to three span-reduce groups, empty and full lists.
usage: fuzz_sparse.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashe_amd.engine import Engine
from oracle import flashe_oracle as orc

KEY = bytes(range(32))


def rand_limbs(rng, n, b):
    L = 2 if b > 64 else 1
    x = rng.integers(0, 2 ** 64, (n, L), dtype=np.uint64)
    if b < 64:
        x[:, 0] &= np.uint64((1 << b) - 1)
    elif 64 < b < 128:
        x[:, 1] &= np.uint64((1 << (b - 64)) - 1)
    return x


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
    engines = {}
    for case in range(cases):
        b = int(rng.choice([128, 128, 100, 64, 23]))
        total = int(rng.choice([1, 5, 4095, 4096, 4097, 8192, 8193, 12_289, 50_000, int(rng.integers(1, 400_000))]))
        C = int(rng.choice([1, 2, 3, 7, 50, 64, 65, 130]))
        if total * C > 6_000_000:
            C = max(1, 6_000_000 // total)
        L = 2 if b > 64 else 1
        eng = engines.setdefault(b, Engine(KEY, b))
        dens = rng.choice([0.0, 0.01, 0.3, 1.0], size=C)
        ks = [int(min(total, max(0, round(d * total + rng.integers(0, 3))))) for d in dens]
        locs = [np.sort(rng.choice(total, size=k, replace=False)).astype(np.uint32) for k in ks]
        vals = [rand_limbs(rng, max(k, 1), b)[:k] for k in ks]
        zeros = [rand_limbs(rng, 1, b)[0] for _ in range(C)]
        dense = [orc.expand_to_dense(total, locs[c], vals[c], zeros[c], b) for c in range(C)]
        want = orc.aggregate_elem(dense, b)
        dl = [eng.upload(l if len(l) else np.zeros(1, dtype=np.uint32)) for l in locs]
        dv = [eng.upload(v if len(v) else np.zeros((1, L), dtype=np.uint64)) for v in vals]
        out = eng.alloc_vec(total)
        tag = (case, b, total, C, ks[:6])
        for srt in (True, False):
            out.upload(np.full(total * L, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64))
            eng.sparse_aggregate_dev(total, dl, ks, dv, zeros, out, sorted_lists=srt)
            assert np.array_equal(out.download(np.uint64, total * L).reshape(total, L), want), ("aggregate", srt) + tag
        it, J = int(rng.integers(0, 1000)), int(rng.choice([1, 4, 16]))
        mm = orc.sparse_minus_mask(KEY, it, locs, total, J, b)
        agg_in = rand_limbs(rng, total, b)
        d_agg = eng.upload(agg_in)
        want_dec = orc.combine(b, agg_in, None, mm)
        for srt in (True, False):
            eng.sparse_minus_mask_dev(it, dl, ks, total, J, out, sorted_lists=srt)
            assert np.array_equal(out.download(np.uint64, total * L).reshape(total, L), mm), ("minus mask", srt) + tag
            out.upload(np.full(total * L, 0x5A5A5A5A5A5A5A5A, dtype=np.uint64))
            eng.sparse_decrypt_dev(it, dl, ks, total, J, d_agg, out, sorted_lists=srt)
            assert np.array_equal(out.download(np.uint64, total * L).reshape(total, L), want_dec), ("decrypt", srt) + tag
        eng.sync()
    print("FUZZ_SPARSE_OK", cases, "cases")


if __name__ == "__main__":
    main()
