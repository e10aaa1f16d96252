"""What a hipEventRecord between two launches costs on the stream (bench.py records events inside its timed steps): K small launches
back to back, with 0 / 1 / 2 event records after each."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd import engine as E  # noqa: E402


def main():
    eng = E.Engine(bytes(range(32)), 128, device=0)
    for n in (61_706, 10_000_000):
        a, m, o = eng.upload(np.zeros((n, 2), dtype=np.uint64)), eng.upload(np.ones((n, 2), dtype=np.uint64)), eng.alloc_vec(n)
        K = 200
        evs = [eng.event() for _ in range(2 * K)]
        for per in (0, 1, 2, 0, 1, 2):
            for _ in range(20):
                eng.combine_dev(n, a, 2, m, None, o)
            eng.sync()
            t = time.perf_counter()
            for k in range(K):
                eng.combine_dev(n, a, 2, m, None, o)
                for e in range(per):
                    eng.record(evs[2 * k + e])
            eng.sync()
            print(f"n={n}: {per} event records per launch: {(time.perf_counter() - t) / K * 1e6:.2f} us per launch", flush=True)


if __name__ == "__main__":
    main()
