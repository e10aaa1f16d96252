#!/usr/bin/env python3
"""Two BUILDS of the library alternated inside one process (the shader clock and the box are then the same for both): ten 1e7-element
chained encrypts per bit width, HIP-event times.
usage: ab_two_libs.py <other .so in flashe_amd/> [bits ...]      e.g. after
       hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DFLASHE_WALK_LOAD64 -shared -o ../libflashe_hip_ab.so kernels.hip sparsify.hip mt19937.hip abi.hip comm.hip -ldl"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import _lib  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402


def engine_from(name, b):
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "flashe_amd", name)
    return Engine(bytes(range(32)), b)


other = sys.argv[1]
n, C = 10_000_000, 10
for b in [int(v) for v in sys.argv[2:]] or [20, 16, 25, 8]:
    engs = {"libflashe_hip.so": engine_from("libflashe_hip.so", b), other: engine_from(other, b)}
    runs = {}
    for name, eng in engs.items():
        pts = [eng.upload(np.random.default_rng(c).integers(0, 2 ** min(b - 1, 40), n, dtype=np.uint64)) for c in range(C)]
        cts = [eng.alloc_vec(n) for _ in range(C)]
        runs[name] = (eng, (lambda e=eng, p=pts, c=cts: e.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, p, 1, c)), eng.event(), eng.event())
    res = {k: [] for k in engs}
    for rep in range(6):
        for name, (eng, run, e0, e1) in runs.items():
            run(); run()
            eng.record(e0)
            for _ in range(10):
                run()
            eng.record(e1)
            res[name].append(eng.elapsed_ms(e0, e1) / 10)
    print(f"b={b}: " + " | ".join(f"{k} {min(v):.4f} ms (all {[round(x, 3) for x in v]})" for k, v in res.items()), flush=True)
