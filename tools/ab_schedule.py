#!/usr/bin/env python3
"""A/B of the round schedules of flashe_amd.dist.ShardedRound on one GPU (BASELINE config 2), in one process so
that the numbers share a box: run (sequential), run_pipelined(chunks), run_fused(chunks).  Prints ms per round."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flashe_amd.dist import HipOps, ShardedRound  # noqa: E402
from flashe_amd.engine import Engine  # noqa: E402

KEY = bytes(range(32))


def main():
    n, C, b, J = 10_000_000, 10, 128, 16
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream, side_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng = Engine(KEY, b, device=0, stream=stream.cuda_stream)
        side = Engine(KEY, b, device=0, stream=side_stream.cuda_stream)
        rnd = ShardedRound(HipOps(eng, side, side_stream), n, b, C, J, dev)
        host = [np.random.Generator(np.random.PCG64(1000 + c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
        pts = [torch.from_numpy(p.view(np.int64)).to(dev) for p in host]
        lo = np.zeros(n, dtype=np.uint64)
        for p in host:
            lo += p
        variants = [("run", None)] + [(m, q) for q in (2, 4, 8) for m in ("pipe", "fused")]
        for rep in range(2):
            for mode, q in variants:
                fn = {"run": lambda it: rnd.run(it, pts, 1), "pipe": lambda it: rnd.run_pipelined(it, pts, 1, chunks=q),
                      "fused": lambda it: rnd.run_fused(it, pts, 1, chunks=q)}[mode]
                out = fn(0)
                torch.cuda.synchronize()
                assert np.array_equal(out[: 2 * n].cpu().numpy().view(np.uint64).reshape(n, 2)[:, 0], lo), (mode, q)
                for it in range(3):
                    fn(it)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                K = 20
                for it in range(K):
                    fn(it)
                torch.cuda.synchronize()
                print(f"rep {rep} {mode:5s} chunks={q}: {(time.perf_counter() - t0) * 1e3 / K:.4f} ms/round", flush=True)


if __name__ == "__main__":
    main()
