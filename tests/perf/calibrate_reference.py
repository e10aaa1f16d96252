#!/usr/bin/env python3
"""Container-only calibration: time the UNMODIFIED reference (imported with
the shims of tests/golden/gen_golden.py) and the CPU oracle side by side on identical inputs."""
import importlib.util
import os
import sys
import time
from functools import reduce

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tests", "golden", "gen_golden.py"))
gg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gg)
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes(range(32))


def main():
    n, C, b, J = 200_000, 10, 128, os.cpu_count()
    gg.RF.N_JOBS = J
    rng = np.random.Generator(np.random.PCG64(1))
    pt = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
    c = gg.new_cipher(b, "double", 0, 0, C)
    obj = pt.astype(object)
    t0 = time.perf_counter(); ct = c.encrypt(obj); t1 = time.perf_counter()
    models = [ct] * C
    agg = reduce(lambda x, y: (x + y) % (1 << b), models); t2 = time.perf_counter()
    d = gg.new_cipher(b, "double", 0, 0, C)
    d.set_idx_list(raw_idx_list=[0] * C, mode="decrypt")
    dec = d.decrypt(agg); t3 = time.perf_counter()
    ref = {"encrypt_s": t1 - t0, "aggregate_s": t2 - t1, "decrypt_s": t3 - t2}
    orc.mask(KEY, 0, 0, 1000, 1, b)
    t0 = time.perf_counter(); oct_ = orc.encrypt(KEY, 0, 0, "double", J, b, pt); t1 = time.perf_counter()
    oagg = orc.aggregate_elem([oct_] * C, b); t2 = time.perf_counter()
    odec = orc.decrypt(KEY, 0, [1] * C, [0] * C, J, b, oagg); t3 = time.perf_counter()
    assert orc.limbs_to_ints(oct_[:1000]) == [int(v) for v in ct[:1000]]
    assert orc.limbs_to_ints(odec[:1000]) == [int(v) for v in dec[:1000]]
    o = {"encrypt_s": t1 - t0, "aggregate_s": t2 - t1, "decrypt_s": t3 - t2}
    print({"n": n, "C": C, "b": b, "cores": J, "reference_via_shims": ref, "oracle": o,
           "ratio_ref_over_oracle": {k: ref[k] / o[k] for k in ref}})


if __name__ == "__main__":
    main()
