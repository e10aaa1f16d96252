#!/usr/bin/env python3
"""One-off differential fuzz of the chained PRF launches against the oracle: random bit widths, vector lengths, chunkings, job
lists (chains of random length over random ragged ranges, broken runs, with / without input, single / double).
usage: fuzz_chains.py [cases] [seed]"""
import os
import sys

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes(range(32))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
engines = {}
for case in range(cases):
    b = int(rng.choice([128, 128, 127, 100, 65, 64, 64, 57, 43, 42, 40, 33, 32, 26, 25, 23, 20, 8, 1]))
    L = 2 if b > 64 else 1
    n = int(rng.choice([1, 63, 64, 255, 256, 257, 1000, 4097, 61_706, 100_003, 300_000, int(rng.integers(1, 2_000_000))]))
    J = int(rng.choice([1, 3, 7, 16]))
    eng = engines.setdefault(b, Engine(KEY, b))
    dbl = bool(rng.integers(0, 2))
    pt = rng.integers(0, 2 ** min(b, 63), n, dtype=np.uint64)
    dpt = eng.upload(pt)
    ptl = np.zeros((n, L), dtype=np.uint64)
    ptl[:, 0] = pt
    jobs, spec, outs, masks = [], [], [], {}
    for _ in range(int(rng.integers(1, 5))):                       # groups: each a run of consecutive prefixes on one range
        kind = int(rng.integers(0, 3))
        first = 0 if kind == 0 else int(rng.integers(0, n))
        count = n - first if kind != 2 else int(rng.integers(0, n - first + 1))
        run = int(rng.choice([1, 1, 2, 3, 10, 17, 40]))
        base = int(rng.integers(0, 2 ** 32 - 64))
        with_in = bool(rng.integers(0, 2))
        for c in range(run):
            a, m = base + c, base + c + 1
            o = eng.alloc_vec(max(count, 1))
            outs.append(o)
            spec.append((a, m, first, count, with_in))
            jobs.append((a, m if dbl else None, first, count, dpt.ptr + 8 * first if with_in else None, 1, o))
    eng.prf_jobs_dev(5, n, J, jobs)
    for (a, m, first, count, with_in), o in zip(spec, outs):
        if count == 0:
            continue
        for i in (a, m):
            if i not in masks:
                masks[i] = orc.mask(KEY, 5, i, n, J, b)
        z = np.zeros((count, L), dtype=np.uint64)
        want = orc.combine(b, ptl[first:first + count] if with_in else z, masks[a][first:first + count], masks[m][first:first + count] if dbl else z)
        got = o.download(np.uint64, count * L).reshape(count, L)
        assert np.array_equal(got, want), (case, b, n, J, dbl, a, first, count, with_in, len(jobs))
    for o in outs:
        o.free()
    dpt.free()
print(f"FUZZ_OK {cases} cases")
