"""Worker for the -m gpu test of flashe_amd.dist: ONE rank on device 0 with a 1-rank RCCL communicator created through the
C ABI (flashe_rccl_*), force_collectives=True, so the exact N > 1 code path (all-to-all, slice kernels, device-side carry
resolution, all-gather) runs through libflashe_hip.so and RCCL and is compared with the oracle.  With
FLASHE_RCCL_SELF_SENDRECV=1 (set by the test) the rank's own piece travels through grouped ncclSend / ncclRecv too.
No PyTorch anywhere: the process fails if torch gets imported."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flashe_amd.dist import HipOps, RcclComm, ShardedRound  # noqa: E402
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE, Engine  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes(range(32))


def main():
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    comm = None
    # (the 2.2 M-element case is long enough for the one-launch encrypt + partial aggregate of `partial`; shorter vectors take its
    # two-call form)
    for b, n, C, J, scheme in [(128, 100_003, 3, 16, SCHEME_DOUBLE), (128, 2_200_013, 3, 16, SCHEME_DOUBLE), (20, 50_001, 4, 16, SCHEME_DOUBLE),
                               (64, 7777, 2, 4, SCHEME_SINGLE)]:
        L = 2 if b > 64 else 1
        eng, side = Engine(KEY, b, device=0), Engine(KEY, b, device=0)
        if comm is None:
            comm = RcclComm.from_env(eng)
            assert (comm.rank, comm.world) == (0, 1)
            assert comm.allreduce(eng, 3.5, RcclComm.MAX) == 3.5
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
        for force in (True, False):
            ops = HipOps(eng, side, comm if force else None)
            pts = [(ops.upload(p), 0) for p in host]
            rnd = ShardedRound(ops, n, b, C, J, scheme=scheme, force_collectives=force)
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
                assert np.array_equal(got, want_packed if mode == "packed" else want_elem), (b, n, force, mode)
                if mode == "packed":
                    src = rnd.k_full if force else rnd.k_partial
                    assert np.array_equal(ops.read((src, 0), len(agg_packed)), agg_packed), (b, n, force)
            if force and L == 1:
                # int_bits <= 64: RCCL's own all-reduce (ncclAllReduce(uint64, sum) + mask) as the exchange, through the REAL library
                rnd_ar = ShardedRound(ops, n, b, C, J, scheme=scheme, force_collectives=True, collective="allreduce")
                for partial in (False, True):
                    out = rnd_ar.run(4, pts, 1, partial_agg=partial)
                    assert np.array_equal(ops.read((out, 0), n * L).reshape(n, L), want_elem), (b, n, "allreduce", partial)
    comm.barrier(eng)
    comm.close()
    assert "torch" not in sys.modules, "the multi-GPU path must not need PyTorch"
    print("DIST_GPU_OK")


if __name__ == "__main__":
    main()
