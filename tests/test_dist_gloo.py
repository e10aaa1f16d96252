"""CPU-only: the N > 1 path (flashe_amd.dist.ShardedRound: all-to-all reduce-scatter, sliced
decrypt, all-gather) with world_size 2 and 3 over gloo."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

KEY = bytes(range(32))


@pytest.mark.parametrize("world,port", [(2, 29541), (3, 29542)])
def test_sharded_round_gloo(world, port, oracle):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_single_rank_schedules(oracle):
    """ShardedRound.run_pipelined / run_fused / run_packed compute the same round as run() on one rank, with and without the
    (identity) exchange step forced."""
    import numpy as np
    from flashe_amd.dist import ShardedRound, deal_clients
    from oracle_ops import OracleOps
    assert [len(x) for x in deal_clients(10, 8)] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert deal_clients(10, 8)[1] == [2, 3] and deal_clients(3, 1) == [[0, 1, 2]] and deal_clients(2, 4) == [[0], [1], [], []]
    for b, n, cpr, chunks in [(128, 5000, 3, 4), (128, 1000, 2, 8), (64, 3000, 2, 2), (20, 2500, 4, 3)]:
        L = 2 if b > 64 else 1
        pts = [np.random.Generator(np.random.PCG64(50 + c)).integers(0, 2 ** (min(b, 64) - 8), n, dtype=np.uint64) for c in range(cpr)]
        want = np.zeros(n, dtype=np.uint64)
        for p in pts:
            want += p
        for force in (False, True):
            for mode in ("run", "partial", "pipe", "fused", "packed"):
                ops = OracleOps(b)
                refs = [(ops.upload(p), 0) for p in pts]
                rnd = ShardedRound(ops, n, b, cpr, 16, force_collectives=force)
                if mode in ("run", "partial"):
                    res = rnd.run(3, refs, 1, partial_agg=(mode == "partial"))
                elif mode == "packed":
                    res = rnd.run_packed(3, refs, 1)
                else:
                    res = (rnd.run_pipelined if mode == "pipe" else rnd.run_fused)(3, refs, 1, chunks=chunks)
                got = ops.read((res, 0), n * L).reshape(n, L)
                if mode != "packed":
                    assert np.array_equal(got[:, 0], want), (b, n, mode, force)
                else:
                    # the packed reduce lets carries cross element boundaries: compare with the oracle's packed round
                    cts = [oracle.encrypt(KEY, 3, c, "double", 16, b, pts[c]) for c in range(cpr)]
                    agg = oracle.aggregate_packed([oracle.pack(ct, b) for ct in cts], n * b)
                    ref = oracle.combine(b, oracle.unpack(agg, n, b), oracle.mask_sum(KEY, 3, [cpr], n, 16, b), oracle.mask_sum(KEY, 3, [0], n, 16, b))
                    assert np.array_equal(got, ref), (b, n, mode, force)


def test_single_rank_element_sharding(oracle):
    """ShardedRound(shard="elements") on one rank (the slice is the whole vector), with and without the forced (identity) all-gather:
    the same decrypted aggregate as the client-sharded round, element-wise and packed."""
    import numpy as np
    from flashe_amd.dist import ShardedRound
    from oracle_ops import OracleOps
    for b, n, C in [(128, 5000, 3), (64, 3000, 2), (20, 2500, 4), (128, 256, 1)]:
        L = 2 if b > 64 else 1
        pts = [np.random.Generator(np.random.PCG64(70 + c)).integers(0, 2 ** (min(b, 64) - 8), n, dtype=np.uint64) for c in range(C)]
        cts = [oracle.encrypt(KEY, 3, c, "double", 16, b, pts[c]) for c in range(C)]
        want = oracle.decrypt(KEY, 3, [C], [0], 16, b, oracle.aggregate_elem(cts, b))
        wantp = oracle.decrypt(KEY, 3, [C], [0], 16, b, oracle.unpack(oracle.aggregate_packed([oracle.pack(ct, b) for ct in cts], n * b), n, b))
        for force in (False, True):
            ops = OracleOps(b)
            refs = [(ops.upload(p), 0) for p in pts]
            rnd = ShardedRound(ops, n, b, C, 16, force_collectives=force, shard="elements")
            assert rnd.element_range() == (0, n) and rnd.element_range(packed=True) == (0, n)
            for partial in (True, False):
                assert np.array_equal(ops.read((rnd.run(3, refs, 1, partial_agg=partial), 0), n * L).reshape(n, L), want), (b, n, force, partial)
            assert np.array_equal(ops.read((rnd.run_packed(3, refs, 1), 0), n * L).reshape(n, L), wantp), (b, n, force, "packed")


def test_single_rank_sparse_position_sharding(oracle):
    """SparseShardedRound on one rank: the position range is the whole vector, the round trip is the plain sparse sum."""
    import numpy as np
    from flashe_amd.dist import SparseShardedRound
    from oracle_ops import OracleOps
    b, total, C, k = 128, 7_000, 4, 250
    ops = OracleOps(b)
    rnd = SparseShardedRound(ops, total, b, C, 16)
    assert rnd.position_range() == (0, total)
    rng = [np.random.Generator(np.random.PCG64(900 + c)) for c in range(C)]
    locs = [np.sort(r.choice(total, k, replace=False)).astype(np.uint32) for r in rng]
    vals = [r.integers(0, 2 ** 60, k, dtype=np.uint64) for r in rng]
    rl, rp = [(ops.upload(l), 0) for l in locs], [(ops.upload(v), 0) for v in vals]
    rc = [(ops.alloc(2 * k), 0) for _ in range(C)]
    out = rnd.run(2, rl, [k] * C, rp, 1, [5] * C, rc)
    want = np.full(total, np.uint64(5 * C), dtype=np.uint64)
    for c in range(C):
        want[locs[c]] += vals[c] - np.uint64(5)
    res = ops.read((out, 0), 2 * total).reshape(total, 2)
    assert np.array_equal(res[:, 0], want) and not res[:, 1].any()
    for c in range(C):
        assert np.array_equal(ops.read(rc[c], 2 * k).reshape(k, 2), oracle.encrypt(KEY, 2, c, "single", 16, b, vals[c]))


def test_rendezvous_file_hands_the_id_to_every_rank(tmp_path, monkeypatch):
    """The torch-free rendezvous of flashe_amd.dist: rank 0 publishes 128 bytes atomically, the others poll for them."""
    import threading
    from flashe_amd.dist import rendezvous_unique_id
    monkeypatch.setenv("FLASHE_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "12345")
    ident = bytes(range(128))
    got = {}

    def reader(r):
        got[r] = rendezvous_unique_id(r, 3, None, timeout=20)[0]
    ts = [threading.Thread(target=reader, args=(r,)) for r in (1, 2)]
    for t in ts:
        t.start()
    got[0] = rendezvous_unique_id(0, 3, lambda: ident)[0]
    for t in ts:
        t.join()
    assert got == {0: ident, 1: ident, 2: ident}
