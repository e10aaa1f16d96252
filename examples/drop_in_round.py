#!/usr/bin/env python3
"""The reference's round through the drop-in class, three ways (needs an MI355X):

  1. exactly as the reference's callers do it -- 1-D object arrays of Python ints in and out (jzf_flashe_block.py:142-174);
  2. uint64 limb arrays (no Python-int conversion);
  3. DeviceVector handles between encrypt, aggregate and decrypt: every plaintext goes up once, one result comes down.

    python examples/drop_in_round.py [n]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flashe_amd.cipher as fc  # noqa: E402
from flashe_amd import FlasheCipher  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
C, int_bits = 4, 128
fc.N_JOBS = 16                                            # must equal the peers' jzf_flashe.N_JOBS
seed = bytes(range(32))


def client(idx):
    c = FlasheCipher(int_bits)                            # same constructor as federatedml.secureprotol.jzf_flashe.FlasheCipher
    c.set_num_clients(C)
    c.generate_prp_seed(seed)
    c.set_iter_index(0)
    c.idx = idx
    return c


clients = [client(i) for i in range(C)]
plain = [np.random.default_rng(i).integers(0, 2 ** 40, n, dtype=np.uint64) for i in range(C)]
want = sum(p.astype(object) for p in plain)

for name, conv, kw in (("object arrays", lambda p: p.astype(object), {}), ("uint64 limbs", lambda p: p, {}), ("device handles", lambda p: p, {"device": True})):
    pts = [conv(p) for p in plain]
    for rep in range(3):                                  # the first passes pay for device start-up, staging blocks and result arrays
        t0 = time.perf_counter()
        cts = [clients[i].encrypt(pts[i], **kw) for i in range(C)]
        agg = clients[0].aggregate(cts)
        clients[0].set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
        dec = clients[0].decrypt(agg, device=False) if kw else clients[0].decrypt(agg)
        ms = (time.perf_counter() - t0) * 1e3
    got = np.asarray(dec).reshape(n, -1)[:, 0] if np.asarray(dec).dtype != object else dec
    assert [int(v) for v in got[:1000]] == [int(v) for v in want[:1000]]
    print(f"{name:15s}: {ms:8.2f} ms for {C} x encrypt + aggregate + decrypt of {n} elements")
