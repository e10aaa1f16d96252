"""Ten compact-layout encrypts (+ their sum) at a given width and length, compile-time width (1) or run-time width (0), timed by HIP
events per launch: back to back, and alternating with the decrypt of the sum as in bench.py's round.
usage: compact_encrypt_time.py {1|0} int_bits n   (tuning library)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FLASHE_LIB_NAME", "libflashe_hip_tuning.so")
os.environ["FLASHE_SMALL_FIXED"] = sys.argv[1] if len(sys.argv) > 1 else "1"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
from flashe_amd import engine as E  # noqa: E402

KEY = bytes(range(32))
eng = E.Engine(KEY, b, device=0)
C, K = 10, 30
rng = np.random.default_rng(1)
pt = [eng.upload(rng.integers(0, 2 ** b, n, dtype=np.uint64).astype(np.uint32)) for _ in range(C)]
ct = [eng.alloc(4 * n + 16) for _ in range(C)]
ds, dec = eng.alloc(4 * n + 16), eng.alloc(4 * n + 16)
idx = list(range(C))
enc = lambda it: eng.encrypt_batch_u32_dev(it, idx, E.SCHEME_DOUBLE, n, 16, pt, ct)                  # noqa: E731
encsum = lambda it: eng.encrypt_batch_sum_u32_dev(it, idx, E.SCHEME_DOUBLE, n, 16, pt, ct, ds)       # noqa: E731
decr = lambda it: eng.aggregate_decrypt_u32_dev(it, [C], [0], n, 16, 0, n, [ds], None, dec, 4)       # noqa: E731
ev = [[eng.event() for _ in range(3)] for _ in range(K)]


def timed(first, second=None):
    for it in range(8):
        first(it)
        if second:
            second(it)
    for k in range(K):
        eng.record(ev[k][0])
        first(k)
        eng.record(ev[k][1])
        if second:
            second(k)
        eng.record(ev[k][2])
    eng.sync()
    a = np.array([[eng.elapsed_ms(e[0], e[1]), eng.elapsed_ms(e[1], e[2])] for e in ev])
    return a[:, 0].mean(), a[:, 0].min(), a[:, 1].mean()


tag = f"b={b} n={n} {'fixed' if os.environ['FLASHE_SMALL_FIXED'] == '1' else 'run-time'} width:"
for name, f, g in (("encrypts, back to back", enc, None), ("encrypts + sum, back to back", encsum, None), ("encrypts + sum, then the decrypt", encsum, decr),
                   ("encrypts + sum, back to back", encsum, None), ("encrypts + sum, then the decrypt", encsum, decr)):
    m, lo, s = timed(f, g)
    print(tag, f"{name}: {m:.4f} ms (min {lo:.4f})" + (f", second launch {s:.4f}" if g else ""), flush=True)
