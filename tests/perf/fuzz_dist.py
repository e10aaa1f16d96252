#!/usr/bin/env python3
"""One-off fuzz of the sharded round with real kernels: random world sizes (ranks share GPU 0, exchange through tests/shm_comm.py),
vector lengths, bit widths, dealings and chunk counts; every schedule of the client-sharded round and -- round 4 -- the element-sharded round
(element-wise with / without the partial aggregate, and packed) and the sparse round sharded by position ranges against the oracle.  usage: fuzz_dist.py [cases] [seed]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from flashe_amd.dist import HipOps, ShardedRound, SparseShardedRound, deal_clients
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE, Engine
from oracle import flashe_oracle as orc
from shm_comm import ShmComm
KEY = bytes(range(32))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
comm = ShmComm(rank, world, os.environ["FLASHE_TEST_SHM_DIR"])
orc.set_num_threads(1)
for b, n, C, J, dbl, chunks in json.loads(os.environ["CASES"]):
    L = 2 if b > 64 else 1
    scheme = SCHEME_DOUBLE if dbl else SCHEME_SINGLE
    mine = deal_clients(C, world)[rank]
    eng, side = Engine(KEY, b, device=0), Engine(KEY, b, device=0)
    ops = HipOps(eng, side, comm)
    host = [np.random.Generator(np.random.PCG64(7 + c)).integers(0, 2 ** min(b, 63), n, dtype=np.uint64) for c in range(C)]
    name = "double" if dbl else "single"
    cts = [orc.encrypt(KEY, 2, c, name, J, b, host[c]) for c in range(C)]
    if dbl:
        add, minus = orc.mask_sum(KEY, 2, [C], n, J, b), orc.mask_sum(KEY, 2, [0], n, J, b)
    else:
        add, minus = np.zeros((n, L), dtype=np.uint64), orc.mask_sum(KEY, 2, list(range(C)), n, J, b)
    want = orc.combine(b, orc.aggregate_elem(cts, b), add, minus)
    aggp = orc.aggregate_packed([orc.pack(ct, b) for ct in cts], n * b)
    wantp = orc.combine(b, orc.unpack(aggp, n, b), add, minus)
    pts = [(ops.upload(host[c]), 0) for c in mine]
    rnd = ShardedRound(ops, n, b, mine, J, rank=rank, world=world, total_clients=C, scheme=scheme)
    for mode in ("run", "pipe", "fused", "packed"):
        if mode == "fused" and not dbl:
            continue
        out = (rnd.run(2, pts, 1) if mode == "run" else rnd.run_pipelined(2, pts, 1, chunks=chunks) if mode == "pipe"
               else rnd.run_fused(2, pts, 1, chunks=chunks) if mode == "fused" else rnd.run_packed(2, pts, 1))
        got = ops.read((out, 0), n * L).reshape(n, L)
        assert np.array_equal(got, wantp if mode == "packed" else want), (rank, world, b, n, C, J, dbl, chunks, mode)
    # the other partition (round 4): elements instead of clients sharded -- every rank plays every client on its slice
    ernd = ShardedRound(ops, n, b, C, J, rank=rank, world=world, scheme=scheme, shard="elements")
    first, count = ernd.element_range()
    epts = [(ops.upload(host[c][first:first + count]) if count else ops.alloc(2), 0) for c in range(C)]
    for partial in (True, False):
        got = ops.read((ernd.run(2, epts, 1, partial_agg=partial), 0), n * L).reshape(n, L)
        assert np.array_equal(got, want), (rank, world, b, n, C, J, dbl, "elements", partial)
    lo, cnt = ernd.element_range(packed=True)
    ppts = [(ops.upload(host[c][lo:lo + cnt]) if cnt else ops.alloc(2), 0) for c in range(C)]
    got = ops.read((ernd.run_packed(2, ppts, 1), 0), n * L).reshape(n, L)
    assert np.array_equal(got, wantp), (rank, world, b, n, C, J, dbl, "elements packed")
    if b > 64:
        # the sparse round by position ranges (round 4): the same vector length as the dense total, random list lengths per client
        sr = np.random.Generator(np.random.PCG64(n + 31 * C))
        ks = [int(sr.integers(0, min(n, 3000) + 1)) for _ in range(C)]
        locs = [np.sort(sr.choice(n, kc, replace=False)).astype(np.uint32) for kc in ks]
        vals = [sr.integers(0, 2 ** 60, kc, dtype=np.uint64) for kc in ks]
        zs = [int(z) for z in sr.integers(0, 2 ** 31, C)]
        srnd = SparseShardedRound(ops, n, b, C, J, rank=rank, world=world)
        rl = [(ops.upload(l) if l.size else ops.alloc(2), 0) for l in locs]
        rp = [(ops.upload(v) if v.size else ops.alloc(2), 0) for v in vals]
        rc = [(ops.alloc(max(kc, 1) * 2), 0) for kc in ks]
        got = ops.read((srnd.run(3, rl, ks, rp, 1, zs, rc), 0), n * 2).reshape(n, 2)
        want_s = np.full(n, np.uint64(sum(zs) & (2 ** 64 - 1)), dtype=np.uint64)
        for c in range(C):
            want_s[locs[c]] += vals[c] - np.uint64(zs[c])
        assert np.array_equal(got[:, 0], want_s), (rank, world, b, n, C, J, "sparse position-sharded")
comm.barrier(eng)
print("OK")
''' % (ROOT, ROOT)


def main():
    import numpy as np
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
    for g in range(cases):
        world = int(rng.choice([2, 3, 4, 5, 8]))
        batch = []
        for _ in range(3):
            b = int(rng.choice([128, 128, 100, 64, 23, 20, 8]))
            n = int(rng.choice([1, 200, 257, 999, 5000, 61_706, int(rng.integers(1, 400_000))]))
            C = int(rng.integers(1, 2 * world + 3))
            batch.append([b, n, C, int(rng.choice([1, 4, 16])), bool(rng.integers(0, 2)), int(rng.integers(1, 7))])
        with tempfile.TemporaryDirectory() as d:
            procs = [subprocess.Popen([sys.executable, "-c", WORKER], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), FLASHE_TEST_SHM_DIR=d, CASES=json.dumps(batch),
                                               OMP_WAIT_POLICY="passive")) for r in range(world)]
            outs = [p.communicate(timeout=600) for p in procs]
        for r, (p, (so, se)) in enumerate(zip(procs, outs)):
            if p.returncode != 0:
                print(f"FAIL world={world} batch={batch} rank={r}\n{se[-2500:]}")
                sys.exit(1)
        print(f"group {g}: world={world} {batch} ok", flush=True)
    print("FUZZ_DIST_OK")


if __name__ == "__main__":
    main()
