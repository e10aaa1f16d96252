#!/usr/bin/env python3
"""Client.sparsify over a whole model (jzf_aggregator.py:585-613): layer by layer (flashe_sparsify per layer: ~12 launches and
three synchronous transfers each) against every layer at once (flashe_sparsify_batch).  ResNet-50-like: 161 layers, 25.5 M float32
parameters; LeNet: 5 layers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

eng = Engine(bytes(32), 128)
rng = np.random.Generator(np.random.PCG64(0))
resnet = [9408] + [64] * 4 + [s for s in (4096, 16384, 36864, 65536, 147456, 262144, 589824, 1048576, 2359296) for _ in range(6)] + \
         [256, 512, 1024, 2048] * 25 + [2048000, 1000]
lenet = [150, 6, 2400, 16, 48000, 120, 10080, 84, 840, 10]
for name, sizes in (("ResNet-50-like", resnet), ("LeNet-5", lenet)):
    layers = [(rng.standard_normal(s) * 0.05).astype(np.float32) for s in sizes]
    ks = [max(1, s // 100) for s in sizes]
    res = [np.zeros(s, dtype=np.float32) for s in sizes]
    for label, fn in (("layer by layer", lambda: [eng.sparsify(l, k, r) for l, k, r in zip(layers, ks, res)]),
                      ("all layers at once", lambda: eng.sparsify_batch(layers, ks, res))):
        fn()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            out = fn()
            best = min(best, time.perf_counter() - t0)
        print(f"{name}: {len(sizes)} layers, {sum(sizes)} parameters, host arrays in and out, {label}: {best * 1e3:8.2f} ms", flush=True)
    # the dict-level mirror of Client.sparsify: layers copied straight into one device buffer, residuals kept in HBM between rounds
    from flashe_amd import weights as wz
    sp = wz.Sparsifier(0.01)
    best = 1e9
    for _ in range(4):
        w = {f"l{i:03d}": l for i, l in enumerate(layers)}
        t0 = time.perf_counter()
        sp.sparsify(w)
        best = min(best, time.perf_counter() - t0)
    print(f"{name}: Sparsifier.sparsify (dict in, compact layers + packed locations out, residuals on the device): {best * 1e3:8.2f} ms", flush=True)
    # device resident: the kernels alone
    flat = np.concatenate(layers)
    dx, dr = eng.upload(flat), eng.upload(np.zeros_like(flat))
    dl, dv = eng.alloc(4 * sum(ks) + 16), eng.alloc(4 * sum(ks) + 16)
    e0, e1 = eng.event(), eng.event()
    run = lambda: eng.sparsify_batch_dev(sizes, ks, dx, False, dr, dl, dv)
    run()
    best = 1e9
    for _ in range(5):
        eng.record(e0); run(); eng.record(e1); eng.sync()
        best = min(best, eng.elapsed_ms(e0, e1))
    print(f"{name}: device resident, all layers at once: {best:8.3f} ms", flush=True)
