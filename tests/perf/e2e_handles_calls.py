#!/usr/bin/env python3
"""Where the device-handle round of bench.py (e2e_ms_device_handles: ten FlasheCipher.encrypt(host, device=True), aggregate, decrypt to
the host; n = 1e7, b = 128) goes: host time of every call of the steady round, beside ten bare uploads of the same plaintexts and the
bare download of the result."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import cipher as cm  # noqa: E402

n, C, b = int(os.environ.get("E2E_N", 10_000_000)), 10, 128
cm.N_JOBS = 16
pts = [np.random.Generator(np.random.PCG64(c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
clients = []
for c in range(C):
    ci = cm.FlasheCipher(b)
    ci.set_num_clients(C)
    ci.generate_prp_seed(bytes(range(32)))
    ci.set_iter_index(0)
    ci.idx = c
    clients.append(ci)

now = time.perf_counter
for rnd in range(4):
    marks = [now()]
    handles = []
    for c in range(C):
        handles.append(clients[c].encrypt(pts[c], device=True))
        marks.append(now())
    agg = clients[0].aggregate(handles)
    marks.append(now())
    clients[0].set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
    dec = clients[0].decrypt(agg, device=False)
    marks.append(now())
    d = np.diff(marks) * 1e3
    print(f"round {rnd}: total {1e3 * (marks[-1] - marks[0]):6.2f} ms | encrypts " + " ".join(f"{x:.2f}" for x in d[:C])
          + f" | aggregate {d[C]:.2f} | decrypt + download {d[C + 1]:.2f}")
    del handles, agg, dec

eng = clients[0]._engine
bufs = [eng.alloc(8 * n) for _ in range(C)]
for _ in range(3):
    t0 = now()
    for c in range(C):
        bufs[c].upload(pts[c])
    t1 = now()
print(f"ten bare uploads of 80 MB: {1e3 * (t1 - t0):.2f} ms ({C * 8 * n / (t1 - t0) / 1e9:.1f} GB/s)")
out = eng.alloc(16 * n)
for _ in range(3):
    t0 = now()
    r = out.download(np.uint64, 2 * n)
    t1 = now()
    del r
print(f"bare download of 160 MB: {1e3 * (t1 - t0):.2f} ms")
