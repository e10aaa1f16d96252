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
