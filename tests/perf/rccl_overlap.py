#!/usr/bin/env python3
"""Can an RCCL transfer run BESIDE a PRF launch?  One GPU, a 1-rank communicator with FLASHE_RCCL_SELF_SENDRECV=1 (the own piece
goes through RCCL's send / recv kernel): ten-client chained encrypt on the main stream, a 140 MB self send / recv (and an
all-gather) on the side stream, alone and together, with the PRF launch filling all CUs or leaving some free."""
import os
import sys
import time

os.environ["FLASHE_RCCL_SELF_SENDRECV"] = "1"
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("RANK", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.dist import make_hip_ops

n, C, J = 10_000_000, 10, 16
ops = make_hip_ops(bytes(range(32)), 128, 0, two_streams=True, with_comm=True)
eng, side = ops.engine, ops.side
pts = [eng.alloc(n * 8) for _ in range(C)]
cts = [eng.alloc(n * 16) for _ in range(C)]
send, recv = eng.alloc(140_000_000), eng.alloc(140_000_000)
evm = [eng.event() for _ in range(2)]
evs = [side.event() for _ in range(2)]


def enc():
    eng.prf_jobs_dev(1, n, J, [(c, c + 1, 0, n, pts[c].ptr, 1, cts[c].ptr) for c in range(C)])


def xfer():
    ops.comm.all_to_all(side, send.ptr, 140_000_000, recv.ptr, 140_000_000, 140_000_000)


def run(do_enc, do_x, reps=6):
    best = (1e9, 0, 0)
    for _ in range(reps):
        eng.sync(); side.sync()
        t0 = time.perf_counter()
        if do_enc:
            eng.record(evm[0]); enc(); eng.record(evm[1])
        if do_x:
            side.record(evs[0]); xfer(); side.record(evs[1])
        eng.sync(); side.sync()
        wall = (time.perf_counter() - t0) * 1e3
        best = min(best, (wall, eng.elapsed_ms(*evm) if do_enc else 0.0, side.elapsed_ms(*evs) if do_x else 0.0))
    return best


for _ in range(20):
    enc()
print("cus", eng.cu_count)
for limit in (0, eng.cu_count - 8, eng.cu_count - 16, eng.cu_count - 32):
    eng.set_cu_limit(limit)
    for _ in range(10):
        enc()
    a = run(True, False)
    b = run(False, True)
    c = run(True, True)
    print(f"PRF on {limit or eng.cu_count:3d} CUs: encrypt alone {a[1]:.3f} ms | transfer alone {b[2]:.3f} ms | together: wall {c[0]:.3f} ms "
          f"(encrypt {c[1]:.3f}, transfer {c[2]:.3f})", flush=True)
