"""Worker for the -m gpu test of BASELINE config 4 at FULL size with its 8-way partition, both ways, real kernels, eight ranks sharing
this one GPU (tests/shm_comm.py stands in for RCCL): (i) client sharding as north_star deals it -- 10 clients as 2, 2, 1, 1, 1, 1, 1, 1 --
and (ii) element sharding -- every rank plays all ten clients on its eighth of the 25,557,032 elements.  Checked: the decrypted
aggregate == the plaintext sum (size-independent property), and the first and last client's ciphertext (slice) == the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from flashe_amd.dist import HipOps, ShardedRound, deal_clients  # noqa: E402
from flashe_amd.engine import Engine  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402
from shm_comm import ShmComm  # noqa: E402

KEY = bytes(range(32))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n = int(os.environ.get("CONFIG4_N", 25_557_032))
    comm = ShmComm(rank, world, os.environ["FLASHE_TEST_SHM_DIR"])
    orc.set_num_threads(2)
    b, C, J, it = 128, 10, 16, 1
    eng = Engine(KEY, b, device=0)
    ops = HipOps(eng, None, comm)
    lo = np.zeros(n, dtype=np.uint64)
    hi = np.zeros(n, dtype=np.uint64)
    deal = deal_clients(C, world)
    mine = deal[rank]
    assert world != 8 or [len(x) for x in deal] == [2, 2, 1, 1, 1, 1, 1, 1]
    # (i) client sharding
    rnd = ShardedRound(ops, n, b, mine, J, rank=rank, world=world, total_clients=C)
    ernd = ShardedRound(ops, n, b, C, J, rank=rank, world=world, shard="elements")
    first, count = ernd.element_range()
    pts, epts, want_first, want_last = [], [], None, None
    for c in range(C):
        p = np.random.Generator(np.random.PCG64(1000 + c)).integers(0, 2 ** 64, n, dtype=np.uint64)
        new = lo + p
        hi += (new < lo).astype(np.uint64)
        lo = new
        if c in mine:
            pts.append((ops.upload(p), 0))
        epts.append((ops.upload(p[first:first + count]) if count else ops.alloc(2), 0))
        if c in (0, C - 1) and count:
            w = orc.encrypt(KEY, it, c, "double", J, b, p)[first:first + count]
            if c == 0:
                want_first = w
            else:
                want_last = w
    for partial in (True, False):
        got = ops.read((rnd.run(it, pts, 1, partial_agg=partial), 0), 2 * n).reshape(n, 2)
        assert np.array_equal(got[:, 0], lo) and np.array_equal(got[:, 1], hi), (rank, "client sharding", partial)
    # (ii) element sharding
    for partial in (True, False):
        got = ops.read((ernd.run(it, epts, 1, partial_agg=partial), 0), 2 * n).reshape(n, 2)
        assert np.array_equal(got[:, 0], lo) and np.array_equal(got[:, 1], hi), (rank, "element sharding", partial)
    if count:
        assert np.array_equal(ops.read(ernd.ct[0], 2 * count).reshape(count, 2), want_first), (rank, "ct slice of client 0")
        assert np.array_equal(ops.read(ernd.ct[C - 1], 2 * count).reshape(count, 2), want_last), (rank, "ct slice of the last client")
    own = ops.read((ernd.run_elements(it, epts, 1, gather=False), 0), 2 * count).reshape(count, 2)
    assert np.array_equal(own[:, 0], lo[first:first + count]) and np.array_equal(own[:, 1], hi[first:first + count])
    comm.barrier(eng)
    if rank == 0:
        print("CONFIG4_8WAY_OK")


if __name__ == "__main__":
    main()
