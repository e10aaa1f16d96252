#!/usr/bin/env python3
"""The round at int_bits <= 32 in the ABI's one-limb layout (uint64 per element) against the compact layout (uint32 per element):
ten 1e7-element clients, batched encrypt + reduce fused with the decrypt, HIP-event times alternated in one process."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

n, C, J = 10_000_000, 10, 16
for b in [int(v) for v in sys.argv[1:]] or [20, 16, 23, 32, 8]:
    eng = Engine(bytes(range(32)), b)
    host = [np.random.default_rng(c).integers(0, 2 ** min(b, 16), n, dtype=np.uint64) for c in range(C)]
    p64 = [eng.upload(h) for h in host]
    c64 = [eng.alloc_vec(n) for _ in range(C)]
    o64 = eng.alloc_vec(n)
    p32 = [eng.upload(h.astype(np.uint32)) for h in host]
    c32 = [eng.alloc(4 * n) for _ in range(C)]
    o32 = eng.alloc(4 * n)
    idx = list(range(C))
    forms = {
        "uint64": (lambda: eng.encrypt_batch_dev(0, idx, SCHEME_DOUBLE, n, J, p64, 1, c64),
                   lambda: eng.aggregate_decrypt_range_dev(0, [C], [0], n, J, 0, n, c64, None, o64)),
        "uint32": (lambda: eng.encrypt_batch_u32_dev(0, idx, SCHEME_DOUBLE, n, J, p32, c32),
                   lambda: eng.aggregate_decrypt_u32_dev(0, [C], [0], n, J, 0, n, c32, None, o32, 4)),
    }
    e = [eng.event() for _ in range(3)]
    res = {k: [] for k in forms}
    for rep in range(6):
        for name, (enc, dec) in forms.items():
            for _ in range(3):
                enc(); dec()
            t_enc = t_dec = 0.0
            for _ in range(10):
                eng.record(e[0]); enc(); eng.record(e[1]); dec(); eng.record(e[2])
                eng.sync()
                t_enc += eng.elapsed_ms(e[0], e[1]); t_dec += eng.elapsed_ms(e[1], e[2])
            res[name].append((t_enc / 10, t_dec / 10))
    a = o64.download(np.uint64, n)
    bb = o32.download(np.uint32, n)
    assert np.array_equal(a.astype(np.uint32), bb)
    line = []
    for name, v in res.items():
        te, td = min(x[0] for x in v), min(x[1] for x in v)
        line.append(f"{name}: encrypt x{C} {te:.4f} + reduce/decrypt {td:.4f} = {te + td:.4f} ms")
    print(f"b={b}: " + " | ".join(line), flush=True)
