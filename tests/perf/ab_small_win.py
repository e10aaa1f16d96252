#!/usr/bin/env python3
"""int_bits <= 25: the block-numbered tiling (FLASHE_SMALL_WIN=0) against the half-window tiling (=1) of the chained launch, alternated
inside ONE process (the knob is read per launch), ten 1e7-element vectors, HIP-event times.  usage: ab_small_win.py [bits ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

n, C = 10_000_000, 10
for b in [int(v) for v in sys.argv[1:]] or [20, 16, 25, 23, 8]:
    eng = Engine(bytes(range(32)), b)
    pts = [eng.upload(np.random.default_rng(c).integers(0, 2 ** min(b - 1, 40), n, dtype=np.uint64)) for c in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    run = lambda: eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, pts, 1, cts)
    e0, e1 = eng.event(), eng.event()
    res = {"0": [], "1": []}
    for rep in range(6):
        for w in ("0", "1"):
            os.environ["FLASHE_SMALL_WIN"] = w
            run(); run()
            eng.record(e0)
            for _ in range(10):
                run()
            eng.record(e1)
            res[w].append(eng.elapsed_ms(e0, e1) / 10)
    print(f"b={b}: block-numbered {min(res['0']):.4f} ms (all {[round(v, 3) for v in res['0']]}) | half windows {min(res['1']):.4f} ms "
          f"(all {[round(v, 3) for v in res['1']]})", flush=True)
