"""Worker for the -m gpu test of the N > 1 path with REAL kernels and several ranks on one GPU: flashe_amd.dist.ShardedRound +
HipOps exactly as production runs them, the exchange through tests/shm_comm.py instead of RCCL (one-GPU boxes cannot host an RCCL
group of several ranks).  Every schedule, equal and unequal dealing, results against the oracle.  No PyTorch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from flashe_amd.dist import HipOps, ShardedRound, SparseShardedRound, deal_clients  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE, Engine  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402
from shm_comm import ShmComm  # noqa: E402

KEY = bytes(range(32))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = ShmComm(rank, world, os.environ["FLASHE_TEST_SHM_DIR"])
    orc.set_num_threads(2)
    cases = [(128, 300_007, "equal2", 16, SCHEME_DOUBLE), (128, 70_001, "uneven", 16, SCHEME_DOUBLE), (20, 50_001, "equal2", 16, SCHEME_DOUBLE),
             (64, 7777, "uneven", 4, SCHEME_SINGLE), (128, 999, "sparse", 1, SCHEME_DOUBLE)]
    for b, n, dealing, J, scheme in cases:
        L = 2 if b > 64 else 1
        if dealing == "equal2":
            C, mine = 2 * world, list(range(2 * rank, 2 * rank + 2))
        elif dealing == "uneven":
            C = world + 2
            mine = deal_clients(C, world)[rank]
        else:
            C = world - 1
            mine = deal_clients(C, world)[rank]
        eng, side = Engine(KEY, b, device=0), Engine(KEY, b, device=0)
        ops = HipOps(eng, side, comm)
        host = [np.random.Generator(np.random.PCG64(70 + c)).integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for c in range(C)]
        name = "double" if scheme == SCHEME_DOUBLE else "single"
        cts = [orc.encrypt(KEY, 4, c, name, J, b, host[c]) for c in range(C)]
        if scheme == SCHEME_DOUBLE:
            add, minus = orc.mask_sum(KEY, 4, [C], n, J, b), orc.mask_sum(KEY, 4, [0], n, J, b)
        else:
            add, minus = np.zeros((n, L), dtype=np.uint64), orc.mask_sum(KEY, 4, list(range(C)), n, J, b)
        want_elem = orc.combine(b, orc.aggregate_elem(cts, b), add, minus)
        agg_packed = orc.aggregate_packed([orc.pack(ct, b) for ct in cts], n * b)
        want_packed = orc.combine(b, orc.unpack(agg_packed, n, b), add, minus)
        pts = [(ops.upload(host[c]), 0) for c in mine]
        rnd = ShardedRound(ops, n, b, mine, J, rank=rank, world=world, total_clients=C, scheme=scheme)
        for mode in ("run", "partial", "pipe", "fused", "packed"):
            if mode == "fused" and scheme != SCHEME_DOUBLE:
                continue
            if mode in ("run", "partial"):
                out = rnd.run(4, pts, 1, partial_agg=(mode == "partial"))
            elif mode == "pipe":
                out = rnd.run_pipelined(4, pts, 1, chunks=3)
            elif mode == "fused":
                out = rnd.run_fused(4, pts, 1, chunks=3)
            else:
                out = rnd.run_packed(4, pts, 1)
            got = ops.read((out, 0), n * L).reshape(n, L)
            assert np.array_equal(got, want_packed if mode == "packed" else want_elem), (rank, b, n, dealing, mode)
            if mode == "packed":
                assert np.array_equal(ops.read((rnd.k_full, 0), len(agg_packed)), agg_packed), (rank, b, n, dealing)
        if L == 1:                                                          # int_bits <= 64: the all-reduce form of the exchange
            rnd_ar = ShardedRound(ops, n, b, mine, J, rank=rank, world=world, total_clients=C, scheme=scheme, collective="allreduce")
            for partial in (False, True):
                out = rnd_ar.run(4, pts, 1, partial_agg=partial)
                assert np.array_equal(ops.read((out, 0), n * L).reshape(n, L), want_elem), (rank, b, n, dealing, "allreduce", partial)
        for c, ref in zip(mine, rnd.ct):                                   # this rank's own ciphertexts (chained launch) vs the oracle
            assert np.array_equal(ops.read(ref, n * L).reshape(n, L), cts[c]), (rank, b, c)
    # ---- element sharding: every rank plays EVERY client on its own slice of the vectors; real kernels on ranges with
    # non-zero `first`, the chained launch with the slice of the partial aggregate, the backward carry walk of the packed reduce
    for b, n, C, J, scheme in [(128, 300_007, 10, 16, SCHEME_DOUBLE), (128, 70_001, 3, 16, SCHEME_DOUBLE), (20, 50_001, 4, 16, SCHEME_DOUBLE),
                               (64, 7777, 3, 4, SCHEME_SINGLE), (128, 999, 2, 1, SCHEME_DOUBLE), (33, 2_100_003, 2, 8, SCHEME_DOUBLE),
                               # 80 clients on every rank's slice: what `bench.py --gpus 8` (config 2, ten clients per GPU) runs in its element-sharded phase
                               (128, 50_001, 80, 16, SCHEME_DOUBLE), (20, 30_001, 130, 16, SCHEME_DOUBLE)]:
        L = 2 if b > 64 else 1
        eng = Engine(KEY, b, device=0)
        ops = HipOps(eng, None, comm)
        host = [np.random.Generator(np.random.PCG64(170 + c)).integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for c in range(C)]
        name = "double" if scheme == SCHEME_DOUBLE else "single"
        cts = [orc.encrypt(KEY, 4, c, name, J, b, host[c]) for c in range(C)]
        if scheme == SCHEME_DOUBLE:
            add, minus = orc.mask_sum(KEY, 4, [C], n, J, b), orc.mask_sum(KEY, 4, [0], n, J, b)
        else:
            add, minus = np.zeros((n, L), dtype=np.uint64), orc.mask_sum(KEY, 4, list(range(C)), n, J, b)
        want_elem = orc.combine(b, orc.aggregate_elem(cts, b), add, minus)
        want_packed = orc.combine(b, orc.unpack(orc.aggregate_packed([orc.pack(ct, b) for ct in cts], n * b), n, b), add, minus)
        rnd = ShardedRound(ops, n, b, C, J, rank=rank, world=world, scheme=scheme, shard="elements")
        first, count = rnd.element_range()
        pts = [(ops.upload(host[c][first:first + count]) if count else ops.alloc(2), 0) for c in range(C)]
        for partial in (True, False):
            out = rnd.run(4, pts, 1, partial_agg=partial)
            assert np.array_equal(ops.read((out, 0), n * L).reshape(n, L), want_elem), (rank, b, n, C, "elements", partial)
            for c in range(C):
                if count:
                    assert np.array_equal(ops.read(rnd.ct[c], count * L).reshape(count, L), cts[c][first:first + count]), (rank, b, c, "elements ct")
        lo, cnt = rnd.element_range(packed=True)
        ptsp = [(ops.upload(host[c][lo:lo + cnt]) if cnt else ops.alloc(2), 0) for c in range(C)]
        out = rnd.run_packed(4, ptsp, 1)
        assert np.array_equal(ops.read((out, 0), n * L).reshape(n, L), want_packed), (rank, b, n, C, "elements packed")
    # ---- the sparse round sharded by POSITION ranges (SparseShardedRound): every rank plays every client on the spans it owns, real kernels
    # (flashe_sparse_encrypt_aggregate_range_dev / flashe_sparse_decrypt_range_dev), the decrypted ranges all-gathered
    for b, total, C, k, J in [(128, 300_007, 10, 3_000, 16), (100, 70_001, 3, 700, 16), (128, 1_752 * 2 + 9, 4, 200, 1), (128, 999, 2, 999, 16),
                              (128, 500_000, 70, 400, 16)]:
        L = 2
        eng = Engine(KEY, b, device=0)
        ops = HipOps(eng, None, comm)
        rng = [np.random.Generator(np.random.PCG64(270 + c)) for c in range(C)]
        ks = [k if c != 1 else max(k // 3, 1) for c in range(C)]
        locs = [np.sort(r.choice(total, kc, replace=False)).astype(np.uint32) for r, kc in zip(rng, ks)]
        vals = [r.integers(0, 2 ** 60, kc, dtype=np.uint64) for r, kc in zip(rng, ks)]
        zeros = [17 + c for c in range(C)]
        rnd = SparseShardedRound(ops, total, b, C, J, rank=rank, world=world)
        first, count = rnd.position_range()
        rl, rp = [(ops.upload(l), 0) for l in locs], [(ops.upload(v), 0) for v in vals]
        rc = [(ops.alloc(max(kc, 1) * L), 0) for kc in ks]
        out = rnd.run(6, rl, ks, rp, 1, zeros, rc)
        want = np.full(total, np.uint64(sum(zeros)), dtype=np.uint64)
        for c in range(C):
            want[locs[c]] += vals[c] - np.uint64(zeros[c])
        got = ops.read((out, 0), total * L).reshape(total, L)
        hi_mask = np.uint64((1 << (b - 64)) - 1) if b < 128 else np.uint64(2 ** 64 - 1)
        assert np.array_equal(got[:, 0], want) and not (got[:, 1] & hi_mask).any(), (rank, b, total, C, "sparse position-sharded")
        for c in range(C):                              # this rank's ciphertext entries = the whole-list encrypt's, the others untouched
            full = orc.encrypt(KEY, 6, c, "single", J, b, vals[c])
            mine = (locs[c] >= first) & (locs[c] < first + count)
            have = ops.read(rc[c], ks[c] * L).reshape(ks[c], L)
            assert np.array_equal(have[mine], full[mine]) and not have[~mine].any(), (rank, b, c, "sparse ct entries")
    # carries rippling through whole element slices of the packed integer
    for b, n in [(128, 256 * world + 5), (64, 300 * world), (20, 256 * world + 17)]:
        L = 2 if b > 64 else 1
        ones = np.full((n, L), np.uint64(2 ** 64 - 1) if b >= 64 else np.uint64(2 ** b - 1), dtype=np.uint64)
        one = np.zeros((n, L), dtype=np.uint64)
        one[n - 1, 0] = 1
        pats = [ones, one]
        eng = Engine(KEY, b, device=0)
        ops = HipOps(eng, None, comm)
        rnd = ShardedRound(ops, n, b, 2, 1, rank=rank, world=world, shard="elements")
        lo, cnt = rnd.element_range(packed=True)
        if cnt:
            for c in range(2):
                sl = np.ascontiguousarray(pats[c][lo:lo + cnt])
                eng._check(eng._lib.flashe_memcpy_h2d(eng._h, ops._a(rnd.ct[c]), sl.ctypes.data, sl.nbytes))
        ops.encrypt_batch_range = lambda *a, **k: None            # keep the planted "ciphertexts"
        rnd.run_elements_packed(0, [(ops.alloc(2), 0)] * 2, 1)
        want = sum(int.from_bytes(orc.pack(p, b).tobytes(), "little") for p in pats) % (1 << (n * b))
        if cnt:
            nl = (cnt * b + 63) // 64
            got = int.from_bytes(ops.read((rnd.ek_sum, 0), nl).tobytes(), "little") & ((1 << (cnt * b)) - 1)
            assert got == (want >> ((n - lo - cnt) * b)) & ((1 << (cnt * b)) - 1), (rank, b, n, "element-sharded carry ripple")
    # adversarial carries through whole limb slices, resolved by the device-side rule
    for b, n in [(128, 64), (64, 37), (20, 500)]:
        L = 2 if b > 64 else 1
        ones = np.full((n, L), np.uint64(2 ** 64 - 1) if b >= 64 else np.uint64(2 ** b - 1), dtype=np.uint64)
        one = np.zeros((n, L), dtype=np.uint64)
        one[n - 1, 0] = 1
        pats = ([ones, one] + [np.zeros_like(one)] * world)[:world]
        eng = Engine(KEY, b, device=0)
        ops = HipOps(eng, None, comm)
        rnd = ShardedRound(ops, n, b, [rank], 1, rank=rank, world=world, total_clients=world)
        rnd._packed_buffers()
        # plant this rank's "ciphertext" and run only the packed reduce part of the round
        eng._check(eng._lib.flashe_memcpy_h2d(eng._h, ops._a(rnd.ct[0]), np.ascontiguousarray(pats[rank]).ctypes.data, pats[rank].nbytes))
        rnd.encrypt_phase = lambda *a, **k: None
        rnd.run_packed(0, [(ops.alloc(n), 0)], 1)
        want = sum(int.from_bytes(orc.pack(p, b).tobytes(), "little") for p in pats) % (1 << (n * b))
        nl = (n * b + 63) // 64
        got = int.from_bytes(ops.read((rnd.k_full, 0), nl).tobytes(), "little")
        assert got == want, (rank, b, n, hex(got)[:40], hex(want)[:40])
    comm.barrier(eng)
    assert "torch" not in sys.modules
    if rank == 0:
        print("DIST_GPU_MULTI_OK")


if __name__ == "__main__":
    main()
