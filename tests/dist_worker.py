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

from flashe_amd.dist import ShardedRound, deal_clients  # noqa: E402
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
    assert ops.allreduce(float(rank), 0) == world - 1 and ops.allreduce(float(rank + 1), 1) == 1.0
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
