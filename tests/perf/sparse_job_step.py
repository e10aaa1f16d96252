#!/usr/bin/env python3
"""The SPARSE job's client step and the arbiter's pass on a ResNet-50-sized model (config 5's job: top-1 % of 25.5 M values, 10 clients here):
Client.sparsify (jzf_aggregator.py:578-623) -> 'zzz' layer -> quantize -> flatten -> strip / encrypt / re-append (:717-743) through
FlasheClient.quantize_encrypt -- fused (one launch over the compact layers, handles out) against call by call (object arrays, the
reference's format) -- then aggregate_sparse_uploads (expand_to_dense + reduce in one pass) and decrypt_unquantize of the dense aggregate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import cipher as cm  # noqa: E402
from flashe_amd.block import FlasheClient, aggregate_sparse_uploads, dynamic_masking_choice  # noqa: E402
from flashe_amd.weights import Sparsifier, from_big_int  # noqa: E402


class W:
    def __init__(self, layers):
        self.walking_order = sorted(layers, key=str)
        self._weights = dict(layers)


cm.N_JOBS = 16
C = 10
rng = np.random.Generator(np.random.PCG64(0))
sizes = [9408] + [s for s in (4096, 16384, 36864, 65536, 147456, 262144, 589824, 1048576, 2359296) for _ in range(6)] + [2048000, 1000]
total = sum(sizes)
args = {"quantize": {"int_bits": 128, "batch": False, "element_bits": 16, "padding": True, "secure": True}, "precompute": {"enable": False},
        "mask": "dynamic"}


_models = {}


def model(c):
    if c not in _models:
        r = np.random.Generator(np.random.PCG64(100 + c))
        _models[c] = {f"l{i:03d}": (r.standard_normal(s) * 0.05).astype(np.float32) for i, s in enumerate(sizes)}
    return {k: v.copy() for k, v in _models[c].items()}


def client(c, fuse):
    cl = FlasheClient(args)
    cl.create_cipher(c, C, bytes(range(32)))
    cl.set_iter_index(1)
    cl.cipher.total = total
    cl.fuse = fuse
    return cl


# 1. every client sparsifies and sends its locations
sparse, masks, t_sp = [], [], 0.0
for c in range(C):
    w = W(model(c))
    sp = Sparsifier(0.01)
    t0 = time.perf_counter()
    enc_loc, le, bits, tot = sp.sparsify(w._weights, w.walking_order)
    if c:                                           # (the first call carries the one-time costs: allocator, tables)
        t_sp += time.perf_counter() - t0
    sparse.append(w)
    masks.append(np.asarray(from_big_int(enc_loc, le, bits, as_object=False)).astype(np.int64).reshape(-1))
print(f"{'client: sparsify (top 1 % of ' + str(total) + ' values, ' + str(len(sizes)) + ' layers)':62s}: {1e3 * t_sp / (C - 1):8.1f} ms")
# 2. the arbiter's hint: the lists go up once (they are needed there for the aggregate anyway), the shared positions are counted on the device
from flashe_amd.engine import Engine  # noqa: E402
arb = Engine(bytes(range(32)), 128, device=0)
t0 = time.perf_counter()
d_masks = [arb.upload(m.astype(np.uint32)) for m in masks]
t1 = time.perf_counter()
choice = dynamic_masking_choice([(d, len(m)) for d, m in zip(d_masks, masks)], total, engine=arb)
t2 = time.perf_counter()
assert choice == dynamic_masking_choice(masks, total)
print(f"{'arbiter: location lists to the device':62s}: {1e3 * (t1 - t0):8.2f} ms")
print(f"{'arbiter: dynamic_masking on the device':62s}: {1e3 * (t2 - t1):8.2f} ms -> {choice} ({1e3 * (time.perf_counter() - t2):.1f} ms with host set intersections)")
# 3. quantise + encrypt of the compact layers ('zzz' appended as the job does)
uploads, clients = [], []
for fuse in (True, False):
    t_enc = 0.0
    for c in range(C if fuse else 2):
        cl = client(c, fuse)
        cl.dynamic_masking(choice, masks)
        w = W({k: np.array(v, copy=True) for k, v in sparse[c]._weights.items()})
        cl.quantizer.set_layer_size_list(w)
        w._weights["zzz"] = np.array([0.0])
        w.walking_order = sorted(w._weights, key=str)
        np.random.seed(c + 1)
        t1 = time.perf_counter()
        out = cl.quantize_encrypt(w, device=fuse)
        cl.cipher.engine.sync()
        if c:
            t_enc += time.perf_counter() - t1
        if fuse:
            uploads.append(out._weights[out.walking_order[0]])
            clients.append(cl)
    n_timed = (C if fuse else 2) - 1
    name = "client: quantize + encrypt, one launch, handle out" if fuse else "client: the same call by call (object arrays, the reference's format)"
    print(f"{name:62s}: {1e3 * t_enc / n_timed:8.2f} ms ({len(masks[0])} values)")

best = 1e9
for rep in range(3):
    t2 = time.perf_counter()
    agg = aggregate_sparse_uploads(arb, uploads, d_masks, total, device=True)
    arb.sync()
    best = min(best, time.perf_counter() - t2)
print(f"{'arbiter: expand_to_dense + reduce of ' + str(C) + ' uploads, one pass':62s}: {1e3 * best:8.2f} ms")
cl = clients[0]
best = 1e9
for rep in range(3):
    cl.set_idx_list(list(range(C)))
    cl.shape_dict = {f"l{i:03d}": (s,) for i, s in enumerate(sizes)}
    t0 = time.perf_counter()
    back = cl.decrypt_unquantize(W({"l000": agg}))
    best = min(best, time.perf_counter() - t0)
print(f"{'client: decrypt_unquantize of the dense aggregate':62s}: {1e3 * best:8.1f} ms")
ref = np.zeros(total)
off = 0
err = 0.0
for i, s in enumerate(sizes):
    got = np.asarray(back._weights[f"l{i:03d}"], dtype=np.float64).reshape(-1)
    want = np.zeros(s)
    for c in range(C):
        m = masks[c]
        sel = m[(m >= off) & (m < off + s)] - off
        want[sel] += _models[c][f"l{i:03d}"][sel]
    err = max(err, float(np.max(np.abs(got - want))) if s else 0.0)
    off += s
print(f"max |decrypted sum - sum of the clients' sparsified layers| = {err:.2e} (quantisation step x clients)")
