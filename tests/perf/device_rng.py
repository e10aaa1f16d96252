#!/usr/bin/env python3
"""np.random.random(n) on the host against flashe_mt19937_random_dev, and the quantiser with either."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import quantize as qz  # noqa: E402
from flashe_amd.engine import Engine  # noqa: E402

eng = Engine(bytes(range(32)), 64)
for n in (1_000_000, 10_000_000, 25_557_032):
    np.random.seed(1)
    t0 = time.perf_counter(); u = np.random.random(n); th = time.perf_counter() - t0
    d = eng.alloc(8 * n)
    eng.numpy_random_dev(1000, d)
    np.random.seed(1)
    t0 = time.perf_counter(); eng.numpy_random_dev(n, d); td = time.perf_counter() - t0
    assert d.download(np.float64, n).tobytes() == u.tobytes()
    x = np.random.Generator(np.random.PCG64(2)).standard_normal(n).astype(np.float32)
    tq = {}
    for flag in ("0", "1"):
        os.environ["FLASHE_DEVICE_RNG"] = flag
        qz._static_quantize_padding_asymmetric(x[:100_000], 2.5, 16, as_object=False)
        t0 = time.perf_counter(); qz._static_quantize_padding_asymmetric(x, 2.5, 16, as_object=False); tq[flag] = time.perf_counter() - t0
    print(f"n={n}: np.random.random {th * 1e3:.1f} ms, device {td * 1e3:.1f} ms ({n / td / 1e9:.2f} G draws/s); "
          f"quantize with host draws {tq['0'] * 1e3:.1f} ms, with device draws {tq['1'] * 1e3:.1f} ms")
