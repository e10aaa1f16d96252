"""CPU-only: the N > 1 path (flashe_amd.dist.ShardedRound: all-to-all reduce-scatter, sliced
decrypt, all-gather) with world_size 2 and 3 over gloo."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("world,port", [(2, 29541), (3, 29542)])
def test_sharded_round_gloo(world, port, oracle):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_pipelined_single_rank_schedule(oracle):
    """ShardedRound.run_pipelined (chunked reduce on a side stream) computes the same round as run()."""
    import numpy as np
    import torch
    from flashe_amd.dist import ShardedRound
    from oracle_ops import OracleOps
    for b, n, cpr, chunks in [(128, 5000, 3, 4), (128, 1000, 2, 8), (64, 3000, 2, 2), (20, 2500, 4, 3)]:
        L = 2 if b > 64 else 1
        pts = [np.random.Generator(np.random.PCG64(50 + c)).integers(0, 2 ** (min(b, 64) - 8), n, dtype=np.uint64) for c in range(cpr)]
        tens = [torch.from_numpy(p.view(np.int64).copy()) for p in pts]
        want = np.zeros(n, dtype=np.uint64)
        for p in pts:
            want += p
        for mode in ("run", "pipe", "fused"):
            rnd = ShardedRound(OracleOps(b), n, b, cpr, 16, "cpu")
            res = rnd.run(3, tens, 1) if mode == "run" else (rnd.run_pipelined if mode == "pipe" else rnd.run_fused)(3, tens, 1, chunks=chunks)
            got = res.numpy().view(np.uint64)[: n * L].reshape(n, L)
            assert np.array_equal(got[:, 0], want), (b, n, mode)
