#!/usr/bin/env python3
"""One launch shape, repeated, for rocprofv3 kernel traces: one_shape.py <int_bits> <n> <clients> [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import SCHEME_DOUBLE, Engine  # noqa: E402

b, n, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
eng = Engine(bytes(range(32)), b)
pts = [eng.upload(np.arange(n, dtype=np.uint64)) for _ in range(C)]
cts = [eng.alloc_vec(n) for _ in range(C)]
for _ in range(reps):
    eng.encrypt_batch_dev(0, list(range(C)), SCHEME_DOUBLE, n, 16, pts, 1, cts)
eng.sync()
