#!/usr/bin/env python3
"""The model-wide sparsifier alone, device resident (for rocprofv3: tools/prof_cmd.sh <tag> tests/perf/sparsify_dev.py): ResNet-50-like,
161 layers, 25.5 M float32 parameters, top 1 % per layer, residuals in HBM; prints the HIP-event time per call."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

eng = Engine(bytes(32), 128)
rng = np.random.Generator(np.random.PCG64(0))
sizes = [9408] + [64] * 4 + [s for s in (4096, 16384, 36864, 65536, 147456, 262144, 589824, 1048576, 2359296) for _ in range(6)] + \
        [256, 512, 1024, 2048] * 25 + [2048000, 1000]
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sizes = [25_557_032]
ks = [max(1, s // 100) for s in sizes]
flat = np.concatenate([(rng.standard_normal(s) * 0.05).astype(np.float32) for s in sizes])
dx, dr = eng.upload(flat), eng.upload(np.zeros_like(flat))
dl, dv = eng.alloc(4 * sum(ks) + 16), eng.alloc(4 * sum(ks) + 16)
e0, e1 = eng.event(), eng.event()
run = lambda: eng.sparsify_batch_dev(sizes, ks, dx, False, dr, dl, dv)      # noqa: E731
for _ in range(3):
    run()
best = 1e9
for _ in range(10):
    eng.record(e0); run(); eng.record(e1); eng.sync()
    best = min(best, eng.elapsed_ms(e0, e1))
print(f"{len(sizes)} layers, {sum(sizes)} float32 values, device resident: {best:.3f} ms per call", flush=True)
