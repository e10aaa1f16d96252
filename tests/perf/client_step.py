#!/usr/bin/env python3
"""A client's step on a ResNet-50-sized model (25.5 M fp32 parameters as 161 layers of realistic sizes): the reference's call chain
quantize(weights) -> weights.encrypted(cipher) through the mirror (object arrays between the two calls, as the reference has them)
against FlasheClient.quantize_encrypt (layer up once, draws on the device, one launch per layer), and the way back."""
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


for name in ("fused, device handles", "fused, host arrays", "two calls (object arrays in between, the reference's format)"):
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
            out = cl.quantize(w)
            for k in out.walking_order:
                out._weights[k] = cl.encrypt(np.asarray(out._weights[k]).reshape(-1))
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

# the way back: the aggregate of C such models (here: C copies of this client's ciphertexts) decrypted and unquantised
cl = client()
w = W({k: v.copy() for k, v in layers.items()})
np.random.seed(1)
enc = cl.quantize_encrypt(w, device=True)
agg = W({k: cl.cipher.aggregate([enc._weights[k]] * C) for k in enc.walking_order})
cl.cipher.set_idx_list(raw_idx_list=list(range(1)) * C, mode="decrypt")
best = 1e9
for rep in range(3):
    a2 = W(dict(agg._weights))
    t0 = time.perf_counter()
    back = cl.decrypt_unquantize(a2)
    best = min(best, time.perf_counter() - t0)
print(f"{'decrypt_unquantize of the aggregate (handles in, float64 layers out)':62s}: {best * 1e3:9.1f} ms", flush=True)
