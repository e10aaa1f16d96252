"""Worker for tests/test_dist_gloo.py: world_size ranks on CPU.  The schedules of flashe_amd.dist.ShardedRound are the code
under test; local arithmetic is done by an ops double backed by the oracle and the exchange goes over gloo (this file lives
under tests/; flashe_amd itself never imports torch)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from flashe_amd.dist import ShardedRound, SparseShardedRound, deal_clients  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402
from oracle_ops import GlooComm, OracleOps  # noqa: E402

KEY = bytes(range(32))


class CraftedOps(OracleOps):
    """Ciphertexts replaced by chosen bit patterns, to drive carries through whole limb slices."""

    def __init__(self, b, patterns, comm):
        super().__init__(b, comm)
        self.patterns = patterns          # global client number -> uint64 array [n, L]

    def encrypt_batch(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts, sum_out=None):
        for i, ct in zip(idx_list, cts):
            self._v(ct, n)[:] = self.patterns[i]


def plain(c, n, bits):
    return np.random.Generator(np.random.PCG64(1000 + c)).integers(0, 2 ** bits, n, dtype=np.uint64)


def result_of(ops, buf, n, L):
    return ops.read((buf, 0), n * L).reshape(n, L)


def packed_rounds(rank, world, comm):
    """ShardedRound.run_packed against the one-process restatement: the packed aggregate (limb for limb) and the decrypted
    result; equal and unequal client counts per rank."""
    for b, n, clients, n_jobs, scheme in [(128, 1000, 2, 8, SCHEME_DOUBLE), (128, 3, 1, 1, SCHEME_DOUBLE), (20, 999, 3, 16, SCHEME_DOUBLE),
                                          (64, 130, 2, 4, SCHEME_SINGLE), (7, 41, 1, 2, SCHEME_DOUBLE),
                                          (128, 777, "uneven", 8, SCHEME_DOUBLE), (23, 500, "uneven", 16, SCHEME_SINGLE)]:
        L = 2 if b > 64 else 1
        if clients == "uneven":
            deal = deal_clients(world + 2, world)               # 2, 2, 1, ... per rank
            C, mine_ids = world + 2, deal[rank]
        else:
            C, mine_ids = world * clients, list(range(rank * clients, (rank + 1) * clients))
        ops = OracleOps(b, comm)
        rnd = ShardedRound(ops, n, b, mine_ids, n_jobs, rank=rank, world=world, total_clients=C, scheme=scheme)
        all_pts = [plain(2000 + c, n, min(b, 64)) for c in range(C)]
        mine = [(ops.upload(all_pts[c]), 0) for c in mine_ids]
        out = rnd.run_packed(9, mine, 1)
        name = "double" if scheme == SCHEME_DOUBLE else "single"
        cts = [orc.encrypt(KEY, 9, c, name, n_jobs, b, all_pts[c]) for c in range(C)]
        agg = orc.aggregate_packed([orc.pack(ct, b) for ct in cts], n * b)
        nl = (n * b + 63) // 64
        src = rnd.k_full if world > 1 else rnd.k_partial
        assert np.array_equal(ops.read((src, 0), nl), agg), (rank, b, n, "packed aggregate")
        if scheme == SCHEME_DOUBLE:
            add, minus = orc.mask_sum(KEY, 9, [C], n, n_jobs, b), orc.mask_sum(KEY, 9, [0], n, n_jobs, b)
        else:
            add, minus = np.zeros((n, L), dtype=np.uint64), orc.mask_sum(KEY, 9, list(range(C)), n, n_jobs, b)
        want = orc.combine(b, orc.unpack(agg, n, b), add, minus)
        assert np.array_equal(result_of(ops, out, n, L), want), (rank, b, n, "packed decrypt")
    # carries that ripple through whole slices (resolved by the device-side rule from the gathered triples)
    for b, n in [(128, 64), (64, 37), (20, 500), (128, 5)]:
        L = 2 if b > 64 else 1
        ones = np.full((n, L), np.uint64(2 ** 64 - 1) if b >= 64 else np.uint64(2 ** b - 1), dtype=np.uint64)
        one = np.zeros((n, L), dtype=np.uint64)
        one[n - 1, 0] = 1                     # the packed integer 1
        big = np.zeros((n, L), dtype=np.uint64)
        big[n // 2, 0] = 3
        for pats in ([ones, one] + [np.zeros_like(one)] * world, [ones] * (world + 1), [ones, one, big, ones, one][: world + 1] + [big] * 2):
            pats = (pats * 2)[:world]
            ops = CraftedOps(b, pats, comm)
            rnd = ShardedRound(ops, n, b, 1, 1, rank=rank, world=world)
            rnd.run_packed(0, [(ops.alloc(n), 0)], 1)
            want = sum(int.from_bytes(orc.pack(p, b).tobytes(), "little") for p in pats) % (1 << (n * b))
            nl = (n * b + 63) // 64
            got = int.from_bytes(ops.read((rnd.k_full, 0), nl).tobytes(), "little")
            assert got == want, (rank, b, n, hex(got)[:40], hex(want)[:40])


def element_sharded_rounds(rank, world, comm):
    """ShardedRound(shard="elements") -- SURVEY.md 8e (i): every rank plays EVERY client on its own element slice; the element-wise
    aggregate needs no exchange (only the optional all-gather of the result), the packed one an all-gather of one carry triple per
    rank.  Against the one-process oracle: the decrypted aggregate, every rank's ciphertext slices, and for the packed reduce the
    carries that cross slice boundaries (crafted ciphertexts: ripples through whole slices)."""
    for b, n, C, n_jobs, scheme in [(128, 1000, 3, 8, SCHEME_DOUBLE), (128, 77, 1, 1, SCHEME_DOUBLE), (20, 999, 4, 16, SCHEME_DOUBLE),
                                    (64, 1300, 2, 4, SCHEME_SINGLE), (23, 700, 5, 16, SCHEME_DOUBLE), (128, 3, 2, 1, SCHEME_DOUBLE),
                                    (7, 600, 3, 2, SCHEME_DOUBLE)]:
        L = 2 if b > 64 else 1
        name = "double" if scheme == SCHEME_DOUBLE else "single"
        ops = OracleOps(b, comm)
        rnd = ShardedRound(ops, n, b, C, n_jobs, rank=rank, world=world, scheme=scheme, shard="elements")
        all_pts = [plain(3000 + c, n, max(1, min(b, 64) - 8)) for c in range(C)]
        cts = [orc.encrypt(KEY, 6, c, name, n_jobs, b, all_pts[c]) for c in range(C)]
        want = np.zeros(n, dtype=np.uint64)
        for p in all_pts:
            want += p
        if b < 64:
            want &= np.uint64((1 << b) - 1)
        first, count = rnd.element_range()
        assert first == min(rank * rnd.slice, n) and count == min(rnd.slice, n - first)
        mine = [(ops.upload(all_pts[c][first:first + count]) if count else ops.alloc(2), 0) for c in range(C)]
        for partial in (True, False):
            out = rnd.run(6, mine, 1, partial_agg=partial)
            res = result_of(ops, out, n, L)
            assert np.array_equal(res[:, 0], want) and (L == 1 or not res[:, 1].any()), (rank, b, n, C, "elements", partial)
            for c in range(C):
                if count:
                    assert np.array_equal(ops.read(rnd.ct[c], count * L).reshape(count, L), cts[c][first:first + count]), (rank, b, c, "ct slice")
        own = result_of(ops, rnd.run_elements(6, mine, 1, gather=False), count, L) if count else np.zeros((0, L), dtype=np.uint64)
        assert np.array_equal(own[:, 0], want[first:first + count])
        # the packed reduce: carries cross element AND slice boundaries
        lo, cnt = rnd.element_range(packed=True)
        assert (n - lo) % 256 == 0 or lo == 0
        assert cnt == 0 or ((n - lo - cnt) * b) % 64 == 0, "a slice must end on a limb boundary of the packed integer"
        minep = [(ops.upload(all_pts[c][lo:lo + cnt]) if cnt else ops.alloc(2), 0) for c in range(C)]
        outp = rnd.run_packed(6, minep, 1)
        agg = orc.aggregate_packed([orc.pack(ct, b) for ct in cts], n * b)
        if scheme == SCHEME_DOUBLE:
            add, minus = orc.mask_sum(KEY, 6, [C], n, n_jobs, b), orc.mask_sum(KEY, 6, [0], n, n_jobs, b)
        else:
            add, minus = np.zeros((n, L), dtype=np.uint64), orc.mask_sum(KEY, 6, list(range(C)), n, n_jobs, b)
        wantp = orc.combine(b, orc.unpack(agg, n, b), add, minus)
        assert np.array_equal(result_of(ops, outp, n, L), wantp), (rank, b, n, C, "elements packed")
    # carries that ripple through whole slices
    for b, n in [(128, 64 * world * 4), (64, 300 * world), (20, 256 * world + 17), (128, 5), (8, 256 * (world - 1) + 1)]:
        L = 2 if b > 64 else 1
        ones = np.full((n, L), np.uint64(2 ** 64 - 1) if b >= 64 else np.uint64(2 ** b - 1), dtype=np.uint64)
        one = np.zeros((n, L), dtype=np.uint64)
        one[n - 1, 0] = 1
        big = np.zeros((n, L), dtype=np.uint64)
        big[n // 2, 0] = 3
        for pats in ([ones, one], [ones] * 3, [ones, one, big, ones, one], [big, one]):
            ops = CraftedRangeOps(b, pats, comm)
            rnd = ShardedRound(ops, n, b, len(pats), 1, rank=rank, world=world, shard="elements")
            lo, cnt = rnd.element_range(packed=True)
            rnd.run_elements_packed(0, [(ops.alloc(max(cnt, 1)), 0)] * len(pats), 1)
            want = sum(int.from_bytes(orc.pack(p, b).tobytes(), "little") for p in pats) % (1 << (n * b))
            # this rank's slice of the packed sum: bits [(n - lo - cnt) b, (n - lo) b) of the whole integer
            if cnt:
                nl = (cnt * b + 63) // 64
                got = int.from_bytes(ops.read((rnd.ek_sum, 0), nl).tobytes(), "little") & ((1 << (cnt * b)) - 1)
                assert got == (want >> ((n - lo - cnt) * b)) & ((1 << (cnt * b)) - 1), (rank, b, n, len(pats), hex(got)[:40])


class CraftedRangeOps(OracleOps):
    """Ciphertext slices replaced by chosen bit patterns (element sharding)."""

    def __init__(self, b, patterns, comm):
        super().__init__(b, comm)
        self.patterns = patterns

    def encrypt_batch_range(self, it, idx_list, scheme, n, n_jobs, first, count, pts, pt_limbs, cts, sum_out=None):
        for i, ct in zip(idx_list, cts):
            self._v(ct, count)[:] = self.patterns[i][first:first + count]


def sparse_position_sharded_rounds(rank, world, comm):
    """SparseShardedRound -- SURVEY.md 8e (i) for the sparse path: every rank plays EVERY client on its own position range of the dense
    vector (whole spans); the gathered result must be the plain sparse sum, every rank's ciphertext entries those of the whole-list
    encrypt, and the ranges must tile the vector."""
    for b, total, C, k, n_jobs in [(128, 20_000, 5, 700, 16), (100, 1_752 * 2 + 5, 3, 60, 1), (128, 900, 4, 900, 16), (128, 1_752 * world, 2, 100, 16)]:
        L = 2
        ops = OracleOps(b, comm)
        rnd = SparseShardedRound(ops, total, b, C, n_jobs, rank=rank, world=world)
        first, count = rnd.position_range()
        assert (count == 0 or first % ops.sparse_span() == 0) and first + count <= total
        rng = [np.random.Generator(np.random.PCG64(500 + c)) for c in range(C)]
        ks = [k if c != 1 else max(k // 3, 1) for c in range(C)]
        locs = [np.sort(r.choice(total, kc, replace=False)).astype(np.uint32) for r, kc in zip(rng, ks)]
        vals = [r.integers(0, 2 ** 60, kc, dtype=np.uint64) for r, kc in zip(rng, ks)]
        zeros = [11 + c for c in range(C)]
        idx = [7 * c + 1 for c in range(C)]

        rl, rp = [(ops.upload(l), 0) for l in locs], [(ops.upload(v), 0) for v in vals]
        rc = [(ops.alloc(max(kc, 1) * L), 0) for kc in ks]
        out = rnd.run(9, rl, ks, rp, 1, zeros, rc, idx=idx)
        want = np.full(total, np.uint64(sum(zeros)), dtype=np.uint64)
        for c in range(C):
            want[locs[c]] += vals[c] - np.uint64(zeros[c])
        # the decrypt subtracts the masks of prefixes 0 .. C-1 (set_idx_list_single's sparse branch): with idx = range(C) the round trip
        # is the plain sum; with other indices compare with the oracle's own aggregate - mask
        cts_full = [orc.encrypt(KEY, 9, idx[c], "single", n_jobs, b, vals[c]) for c in range(C)]
        agg = np.zeros((total, L), dtype=np.uint64)
        for c in range(C):
            agg = orc.aggregate_elem([agg, orc.expand_to_dense(total, locs[c], cts_full[c], np.array([[zeros[c], 0]], dtype=np.uint64), b)], b)
        ref = orc.combine(b, agg, None, orc.sparse_minus_mask(KEY, 9, locs, total, n_jobs, b))
        res = result_of(ops, out, total, L) if world > 1 else None
        if world > 1:
            assert np.array_equal(res, ref), (rank, b, total, C, "gathered")
        own = result_of(ops, rnd.run(9, rl, ks, rp, 1, zeros, rc, idx=idx, gather=False), count, L) if count else np.zeros((0, L), dtype=np.uint64)
        assert np.array_equal(own, ref[first:first + count]), (rank, b, total, C, "own range")
        for c in range(C):
            mine = (locs[c] >= first) & (locs[c] < first + count)
            got = ops.read(rc[c], ks[c] * L).reshape(ks[c], L)
            assert np.array_equal(got[mine], cts_full[c][mine]) and not got[~mine].any(), (rank, c, "ciphertext entries")
        # with the clients' own indices 0 .. C-1 the round trip is the plain sparse sum
        out2 = rnd.run(9, rl, ks, rp, 1, zeros, rc, gather=False)
        own2 = result_of(ops, out2, count, L) if count else np.zeros((0, L), dtype=np.uint64)
        mask_b = np.uint64((1 << 64) - 1)
        assert np.array_equal(own2[:, 0] & mask_b, want[first:first + count]), (rank, b, total, "plain sum")
        # the ranges tile the vector: rank g owns [g S, min(total, (g + 1) S)), S whole spans
        assert first == min(total, rank * rnd.slice) and first + count == min(total, (rank + 1) * rnd.slice) and world * rnd.slice >= total


def main():
    dist.init_process_group("gloo")
    comm = GlooComm()
    rank, world = comm.rank, comm.world
    orc.set_num_threads(1)
    cases = [(128, 1000, 2, 8, SCHEME_DOUBLE), (128, 77, 1, 1, SCHEME_DOUBLE), (20, 999, 3, 16, SCHEME_DOUBLE), (64, 130, 2, 4, SCHEME_SINGLE),
             # unequal client counts per rank: BASELINE config 4 deals 10 clients over 8 GPUs as 2, 2, 1, 1, 1, 1, 1, 1
             (128, 1500, "uneven", 16, SCHEME_DOUBLE), (23, 700, "uneven", 16, SCHEME_DOUBLE), (64, 300, "uneven", 4, SCHEME_SINGLE),
             # fewer clients than ranks: some ranks host nobody and still take part in the exchange
             (128, 600, "sparse", 1, SCHEME_DOUBLE)]
    for b, n, clients, n_jobs, scheme in cases:
        L = 2 if b > 64 else 1
        ops = OracleOps(b, comm)
        if clients == "uneven":
            C = world + 2
            mine_ids = deal_clients(C, world)[rank]
            assert [len(x) for x in deal_clients(10, 8)] == [2, 2, 1, 1, 1, 1, 1, 1]
        elif clients == "sparse":
            C = world - 1
            mine_ids = deal_clients(C, world)[rank]
        else:
            C, mine_ids = world * clients, list(range(rank * clients, (rank + 1) * clients))
        rnd = ShardedRound(ops, n, b, mine_ids, n_jobs, rank=rank, world=world, total_clients=C, scheme=scheme)
        pt_bits = min(b, 64) - 8
        all_pts = [plain(c, n, pt_bits) for c in range(C)]
        mine = [(ops.upload(all_pts[c]), 0) for c in mine_ids]
        want = np.zeros(n, dtype=np.uint64)
        for p in all_pts:
            want += p
        if b < 64:
            want &= np.uint64((1 << b) - 1)
        for mode, chunks in (("run", 0), ("partial", 0), ("pipe", 4), ("pipe", 3), ("fused", 4), ("fused", 1), ("fused", 5)):
            if mode == "fused" and scheme != SCHEME_DOUBLE:
                continue
            out = rnd.run(5, mine, 1, partial_agg=(mode == "partial")) if mode in ("run", "partial") else \
                (rnd.run_pipelined if mode == "pipe" else rnd.run_fused)(5, mine, 1, chunks=chunks)
            res = result_of(ops, out, n, L)
            assert np.array_equal(res[:, 0], want), (rank, b, n, clients, mode, chunks)
            if L == 2:
                assert not res[:, 1].any()
        if L == 1:
            # the same round with RCCL's own all-reduce as the exchange (int_bits <= 64: SURVEY.md section 8e)
            rnd_ar = ShardedRound(ops, n, b, mine_ids, n_jobs, rank=rank, world=world, total_clients=C, scheme=scheme, collective="allreduce")
            for partial in (False, True):
                res = result_of(ops, rnd_ar.run(5, mine, 1, partial_agg=partial), n, L)
                assert np.array_equal(res[:, 0], want), (rank, b, n, clients, "allreduce", partial)
    packed_rounds(rank, world, comm)
    element_sharded_rounds(rank, world, comm)
    sparse_position_sharded_rounds(rank, world, comm)
    assert ops.allreduce(float(rank), 0) == world - 1 and ops.allreduce(float(rank + 1), 1) == 1.0
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
