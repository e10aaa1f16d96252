"""Worker for tests/test_dist_gloo.py: world_size ranks on CPU (gloo).  The exchange logic of
flashe_amd.dist.ShardedRound is the code under test; local arithmetic is done by an ops double
backed by the oracle (this file lives under tests/)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from flashe_amd.dist import ShardedRound  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes(range(32))


from oracle_ops import OracleOps  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc.set_num_threads(1)
    for b, n, cpr, n_jobs, scheme in [(128, 1000, 2, 8, SCHEME_DOUBLE), (128, 77, 1, 1, SCHEME_DOUBLE),
                                      (20, 999, 3, 16, SCHEME_DOUBLE), (64, 130, 2, 4, SCHEME_SINGLE)]:
        L = 2 if b > 64 else 1
        ops = OracleOps(b)
        rnd = ShardedRound(ops, n, b, cpr, n_jobs, "cpu", rank=rank, world=world, scheme=scheme)
        pt_bits = min(b, 64) - 8
        all_pts = [np.random.Generator(np.random.PCG64(1000 + c)).integers(0, 2 ** pt_bits, n, dtype=np.uint64)
                   for c in range(world * cpr)]
        mine = [torch.from_numpy(all_pts[rank * cpr + c].view(np.int64).copy()) for c in range(cpr)]
        want = np.zeros(n, dtype=np.uint64)
        for p in all_pts:
            want += p
        if b < 64:
            want &= np.uint64((1 << b) - 1)
        for mode, chunks in (("run", 0), ("pipe", 4), ("pipe", 3)):
            out = rnd.run(5, mine, 1) if mode == "run" else rnd.run_pipelined(5, mine, 1, chunks=chunks)
            res = out.numpy().view(np.uint64)[: n * L].reshape(n, L)
            assert np.array_equal(res[:, 0], want), (rank, b, n, mode, chunks)
            if L == 2:
                assert not res[:, 1].any()
        # the same round through the oracle as one process: identical ciphertext aggregate
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
