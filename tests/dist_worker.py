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


class CraftedOps(OracleOps):
    """Ciphertexts replaced by chosen bit patterns, to drive carries through whole limb slices."""

    def __init__(self, b, patterns):
        super().__init__(b)
        self.patterns = patterns          # global client number -> uint64 array [n, L]

    def encrypt_batch(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts):
        for i, ct in zip(idx_list, cts):
            self._v(ct, n)[:] = self.patterns[i]


def packed_rounds(rank, world):
    """ShardedRound.run_packed against the one-process restatement: the packed aggregate (limb for limb)
    and the decrypted result."""
    for b, n, cpr, n_jobs, scheme in [(128, 1000, 2, 8, SCHEME_DOUBLE), (128, 3, 1, 1, SCHEME_DOUBLE), (20, 999, 3, 16, SCHEME_DOUBLE),
                                      (64, 130, 2, 4, SCHEME_SINGLE), (7, 41, 1, 2, SCHEME_DOUBLE)]:
        L = 2 if b > 64 else 1
        C = world * cpr
        rnd = ShardedRound(OracleOps(b), n, b, cpr, n_jobs, "cpu", rank=rank, world=world, scheme=scheme)
        all_pts = [np.random.Generator(np.random.PCG64(3000 + c)).integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for c in range(C)]
        mine = [torch.from_numpy(all_pts[rank * cpr + c].view(np.int64).copy()) for c in range(cpr)]
        out = rnd.run_packed(9, mine, 1)
        name = "double" if scheme == SCHEME_DOUBLE else "single"
        cts = [orc.encrypt(KEY, 9, c, name, n_jobs, b, all_pts[c]) for c in range(C)]
        agg = orc.aggregate_packed([orc.pack(ct, b) for ct in cts], n * b)
        nl = (n * b + 63) // 64
        src = rnd.k_full if world > 1 else rnd.k_partial
        assert np.array_equal(src.numpy().view(np.uint64)[:nl], agg), (rank, b, n, "packed aggregate")
        if scheme == SCHEME_DOUBLE:
            add, minus = orc.mask_sum(KEY, 9, [C], n, n_jobs, b), orc.mask_sum(KEY, 9, [0], n, n_jobs, b)
        else:
            add, minus = np.zeros((n, L), dtype=np.uint64), orc.mask_sum(KEY, 9, list(range(C)), n, n_jobs, b)
        want = orc.combine(b, orc.unpack(agg, n, b), add, minus)
        assert np.array_equal(out.numpy().view(np.uint64)[: n * L].reshape(n, L), want), (rank, b, n, "packed decrypt")
    # carries that ripple through whole slices
    for b, n in [(128, 64), (64, 37), (20, 500), (128, 5)]:
        L = 2 if b > 64 else 1
        ones = np.full((n, L), np.uint64(2 ** 64 - 1) if b >= 64 else np.uint64(2 ** b - 1), dtype=np.uint64)
        one = np.zeros((n, L), dtype=np.uint64)
        one[n - 1, 0] = 1                     # the packed integer 1
        big = np.zeros((n, L), dtype=np.uint64)
        big[n // 2, 0] = 3
        for pats in ([ones, one] + [np.zeros_like(one)] * world, [ones] * (world + 1), [ones, one, big, ones, one][: world + 1] + [big] * 2):
            pats = (pats * 2)[:world]
            rnd = ShardedRound(CraftedOps(b, pats), n, b, 1, 1, "cpu", rank=rank, world=world)
            rnd.run_packed(0, [torch.zeros(n, dtype=torch.int64)], 1)
            want = sum(int.from_bytes(orc.pack(p, b).tobytes(), "little") for p in pats) % (1 << (n * b))
            nl = (n * b + 63) // 64
            got = int.from_bytes(rnd.k_full.numpy().view(np.uint64)[:nl].tobytes(), "little")
            assert got == want, (rank, b, n, hex(got)[:40], hex(want)[:40])


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
        for mode, chunks in (("run", 0), ("pipe", 4), ("pipe", 3), ("fused", 4), ("fused", 1), ("fused", 5)):
            if mode == "fused" and scheme != SCHEME_DOUBLE:
                continue
            out = rnd.run(5, mine, 1) if mode == "run" else (rnd.run_pipelined if mode == "pipe" else rnd.run_fused)(5, mine, 1, chunks=chunks)
            res = out.numpy().view(np.uint64)[: n * L].reshape(n, L)
            assert np.array_equal(res[:, 0], want), (rank, b, n, mode, chunks)
            if L == 2:
                assert not res[:, 1].any()
        # the same round through the oracle as one process: identical ciphertext aggregate
    packed_rounds(rank, world)
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
