import os, sys
sys.path.insert(0, '/root/repo')
os.environ["FLASHE_LIB_NAME"] = "libflashe_hip_tuning.so"
import numpy as np
from flashe_amd.engine import Engine
total, C = 25_557_032, 50
k = total // 100
eng = Engine(bytes(range(32)), 128, device=0)
rng = [np.random.Generator(np.random.PCG64(2000 + c)) for c in range(C)]
locs = [np.sort(r.choice(total, k, replace=False)).astype(np.uint32) for r in rng]
vals = [r.integers(0, 2 ** 64, k, dtype=np.uint64) for r in rng]
d_loc = [eng.upload(l) for l in locs]; d_val = [eng.upload(v) for v in vals]
d_ct = [eng.alloc_vec(k) for _ in range(C)]
d_agg, d_dec = eng.alloc_vec(total), eng.alloc_vec(total)
t_loc, t_val, t_ct, t_k = eng.ptr_table(d_loc), eng.ptr_table(d_val), eng.ptr_table(d_ct), eng.u64_table([k] * C)
t_zero, idx = eng.zeros_table([1 << 31] * C), list(range(C))
bounds = eng.span_bounds(total, t_loc, t_k)
ev = [eng.event() for _ in range(3)]
for name, fn in (("enc+agg", lambda it: eng.sparse_encrypt_aggregate_dev(it, idx, t_loc, t_k, t_val, 1, t_zero, total, 16, t_ct, d_agg, bounds=bounds)),
                 ("dec", lambda it: eng.sparse_decrypt_dev(it, t_loc, t_k, total, 16, d_agg, d_dec, sorted_lists=True, bounds=bounds))):
    best = 1e9
    for rep in range(4):
        for it in range(5): fn(it)
        eng.record(ev[0])
        for it in range(10): fn(it)
        eng.record(ev[1])
        try:
            eng.sync()
        except Exception as e:
            pass
        best = min(best, eng.elapsed_ms(ev[0], ev[1]) / 10)
    print(name, "probe", os.environ.get("FLASHE_SPAN_PROBE", "0"), "%.4f ms" % best, flush=True)
