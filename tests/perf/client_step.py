#!/usr/bin/env python3
"""A client's step on a ResNet-50-sized model (25.5 M fp32 parameters as 56 layers of realistic sizes): the reference JOB's call chain
quantize(weights) -> flatten_weights -> weights.encrypted(cipher) (jzf_aggregator.py:721-741) through the mirror call by call (object
arrays between the calls, as the reference has them) against FlasheClient.quantize_encrypt (every layer up once into one flat device
buffer, draws on the device, ONE launch over the flattened model with a per-layer alpha table), and the way back.  Also counts the
kernel launches of the fused step (rocprofv3 --kernel-trace --stats -- python3 tests/perf/client_step.py fused-only)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import cipher as cm  # noqa: E402
from flashe_amd.block import FlasheClient  # noqa: E402


class W:
    def __init__(self, layers):
        self.walking_order = sorted(layers)
        self._weights = dict(layers)


cm.N_JOBS = 16
rng = np.random.Generator(np.random.PCG64(0))
sizes = [9408] + [s for s in (4096, 16384, 36864, 65536, 147456, 262144, 589824, 1048576, 2359296) for _ in range(6)] + [2048000, 1000]
total = sum(sizes)
layers = {f"l{i:03d}": (rng.standard_normal(s) * 0.05).astype(np.float32) for i, s in enumerate(sizes)}
args = {"quantize": {"int_bits": 128, "batch": False, "element_bits": 16, "padding": True, "secure": True}, "precompute": {"enable": False}}
C = 10


def client():
    cl = FlasheClient(args)
    cl.create_cipher(3, C, bytes(range(32)))
    cl.set_iter_index(1)
    return cl


names = ("fused, device handles", "fused, host arrays", "call by call (object arrays in between, the reference's format)")
if "fused-only" in sys.argv:
    names = names[:1]
for name in names:
    cl = client()
    best = 1e9
    for rep in range(3 if name.startswith("fused") else 1):
        w = W({k: v.copy() for k, v in layers.items()})
        np.random.seed(1)
        t0 = time.perf_counter()
        if name.startswith("fused"):
            out = cl.quantize_encrypt(w, device=name.endswith("handles"))
            cl.cipher.engine.sync()
        else:
            out = cl.flatten_weights(cl.quantize(w))
            k0 = out.walking_order[0]
            out._weights[k0] = cl.encrypt(out._weights[k0])
        best = min(best, time.perf_counter() - t0)
        del out
    print(f"{name:62s}: {best * 1e3:9.1f} ms for {len(sizes)} layers, {total} parameters", flush=True)

if "--profile" in sys.argv:                 # where the host time of the fused step goes
    import cProfile
    import pstats
    cl = client()
    w = W({k: v.copy() for k, v in layers.items()})
    np.random.seed(1)
    cl.quantize_encrypt(W({k: v.copy() for k, v in layers.items()}), device=True)
    pr = cProfile.Profile()
    pr.enable()
    out = cl.quantize_encrypt(w, device=True)
    cl.cipher.engine.sync()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)

# the BATCHED job (the paper's main configuration: several quantised values per 128-bit ciphertext element): two launches
bargs = {"quantize": {"int_bits": 128, "batch": True, "element_bits": 16, "padding": True, "secure": True}, "precompute": {"enable": False}}
for name in ("batched, fused, device handles", "batched, call by call (object arrays in between)"):
    cl = FlasheClient(bargs)
    cl.create_cipher(3, C, bytes(range(32)))
    cl.set_iter_index(1)
    cl.fuse = name.startswith("batched, fused")
    best = 1e9
    for rep in range(3 if cl.fuse else 1):
        w = W({k: v.copy() for k, v in layers.items()})
        np.random.seed(1)
        t0 = time.perf_counter()
        out = cl.quantize_encrypt(w, device=cl.fuse)
        cl.cipher.engine.sync()
        best = min(best, time.perf_counter() - t0)
    k0b = out.walking_order[0]
    print(f"{name:62s}: {best * 1e3:9.1f} ms ({len(out._weights[k0b])} ciphertext elements of 6 values)", flush=True)
    if cl.fuse:
        aggb = cl.cipher.aggregate([out._weights[k0b]] * C)
        cl.cipher.set_idx_list(raw_idx_list=list(range(1)) * C, mode="decrypt")
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            backb = cl.decrypt_unquantize(W({k0b: aggb}))
            best = min(best, time.perf_counter() - t0)
        assert all(backb._weights[k].shape == layers[k].shape for k in layers)
        print(f"{'batched, decrypt_unquantize (handle in, float64 layers out)':62s}: {best * 1e3:9.1f} ms", flush=True)

# the way back: the aggregate of the C clients' models (every client its own cipher index, the same weights) decrypted and unquantised --
# a round's decrypt: C distinct prefixes telescope to two mask streams
def client_i(i):
    c = FlasheClient(args)
    c.create_cipher(i, C, bytes(range(32)))
    c.set_iter_index(1)
    return c


clients = [client_i(i) for i in range(C)]
encs = []
for c in clients:
    np.random.seed(1)
    encs.append(c.quantize_encrypt(W({k: v.copy() for k, v in layers.items()}), device=True))
cl = clients[3]
k0 = encs[0].walking_order[0]
agg = cl.cipher.aggregate([e._weights[k0] for e in encs])
cl.cipher.set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
best = 1e9
for rep in range(3):
    a2 = W({k0: agg})
    t0 = time.perf_counter()
    back = cl.decrypt_unquantize(a2)
    best = min(best, time.perf_counter() - t0)
assert sorted(back._weights) == sorted(layers) and all(back._weights[k].shape == layers[k].shape for k in layers)
err = max(float(np.max(np.abs(back._weights[k].reshape(-1) / C - layers[k].astype(np.float64)))) for k in layers)
print(f"max |mean of the C decrypted models - the model| = {err:.2e} (the quantisation step and the clipping at alpha)")
print(f"{'decrypt_unquantize of the aggregate (handle in, float64 layers out)':62s}: {best * 1e3:9.1f} ms", flush=True)
