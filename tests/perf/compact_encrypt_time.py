"""Ten compact-layout encrypts (and encrypts + their sum) at a given width and length, compile-time width (1) or run-time width (0):
usage: compact_encrypt_time.py {1|0} int_bits n   (tuning library)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FLASHE_LIB_NAME"] = "libflashe_hip_tuning.so"
os.environ["FLASHE_SMALL_FIXED"] = sys.argv[1]
b = int(sys.argv[2]); n = int(sys.argv[3])
from flashe_amd import engine as E
KEY = bytes(range(32))
eng = E.Engine(KEY, b, device=0)
C = 10
rng = np.random.default_rng(1)
pt = [eng.upload(rng.integers(0, 2 ** b, n, dtype=np.uint64).astype(np.uint32)) for _ in range(C)]
ct = [eng.alloc(4 * n + 16) for _ in range(C)]
ds = eng.alloc(4 * n + 16)
idx = list(range(C))
for f, name in ((lambda: eng.encrypt_batch_u32_dev(3, idx, E.SCHEME_DOUBLE, n, 16, pt, ct), "enc"), (lambda: eng.encrypt_batch_sum_u32_dev(3, idx, E.SCHEME_DOUBLE, n, 16, pt, ct, ds), "enc+sum")):
    for _ in range(5): f()
    eng.sync(); t = time.perf_counter()
    for _ in range(30): f()
    eng.sync(); print(name, "fixed" if sys.argv[1] == "1" else "runtime", b, n, (time.perf_counter() - t) / 30 * 1e3, "ms")
