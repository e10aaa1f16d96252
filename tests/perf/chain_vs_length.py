"""The chained encrypt launch (ten clients + their partial aggregate, int_bits 128) at growing vector lengths: AES blocks per second
against n -- does a ResNet-50-sized vector (config 4) cost more per element than config 2's 1e7?
usage: chain_vs_length.py [n, comma separated]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import engine as E  # noqa: E402

KEY = bytes(range(32))


def main():
    ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5_000_000, 10_000_000, 15_000_000, 20_000_000, 25_557_032, 40_000_000]
    C, J = 10, 16
    eng = E.Engine(KEY, 128, device=0)
    rng = np.random.default_rng(3)
    for n in ns:
        pt = [eng.upload(rng.integers(0, 2 ** 63, n, dtype=np.uint64)) for _ in range(C)]
        ct = [eng.alloc_vec(n) for _ in range(C)]
        dsum = eng.alloc_vec(n)
        idx = list(range(C))
        f = lambda it: eng.encrypt_batch_sum_dev(it, idx, E.SCHEME_DOUBLE, n, J, pt, 1, ct, dsum)      # noqa: E731
        for it in range(5):
            f(it)
        K = 20
        ev = [eng.event() for _ in range(K + 1)]
        eng.record(ev[0])
        for k in range(K):
            f(k)
            eng.record(ev[k + 1])
        eng.sync()
        ms = np.array([eng.elapsed_ms(ev[k], ev[k + 1]) for k in range(K)])
        blocks = (C + 1) * n
        print(f"n={n:9d}  {ms.mean():.4f} ms (min {ms.min():.4f})  {ms.mean() / n * 1e6:.4f} ms per 1e6 elements  {blocks / ms.mean() / 1e6:.2f} G blocks/s", flush=True)
        del pt, ct, dsum


if __name__ == "__main__":
    main()
