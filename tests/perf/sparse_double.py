#!/usr/bin/env python3
"""Sparse + double mask at BASELINE config-5 size (25.5 M positions, 50 clients, top-1 %): both dense decrypt masks from the location
lists in one call (flashe_sparse_double_masks_dev) against the selector-based form it replaces in FlasheCipher.set_idx_list."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine  # noqa: E402

total, C = 25_557_032, 50
k = total // 100
eng = Engine(bytes(range(32)), 128)
rng = np.random.Generator(np.random.PCG64(0))
hot = np.sort(rng.choice(total, 2 * k, replace=False))                    # the clients agree on most of their top weights
locs = [np.unique(np.concatenate([hot[rng.random(2 * k) < 0.4], rng.choice(total, k // 5, replace=False)])).astype(np.uint32) for _ in range(C)]
dloc = [eng.upload(l) for l in locs]
ks = [len(l) for l in locs]
da, dm = eng.alloc_vec(total), eng.alloc_vec(total)
eng.sparse_double_masks_dev(0, dloc, ks, total, da, dm)
eng.sync()
t0 = time.perf_counter()
for it in range(5):
    eng.sparse_double_masks_dev(it, dloc, ks, total, da, dm)
eng.sync()
new_ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"location lists: {sum(ks)} entries, {new_ms:.2f} ms per call (both masks)")
if "--old" in sys.argv:
    ohs = []
    for l in locs:
        a = np.zeros(total, dtype=np.uint8); a[l] = 1; ohs.append(a)
    minus = [ohs[c] & (1 - ohs[c - 1]) if c > 0 else ohs[c] for c in range(C)]
    t0 = time.perf_counter()
    dsel = [eng.upload(m) for m in minus]
    eng.sparse_dense_mask_dev(0, dsel, total, dm)
    eng.sync()
    print(f"selector form, minus side only, incl. uploading {C} x {total} selector bytes: {(time.perf_counter() - t0) * 1e3:.1f} ms")
