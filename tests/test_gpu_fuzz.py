"""Short, seeded runs of the differential fuzzers under tests/perf/ inside the GPU gate: random job lists of chained PRF launches
(every bit-width class: one element per block, direct outputs for 2 .. 4 elements per block, the fast and the general walk) and
random sparse rounds (span bounds / span reduce / fused sparse decrypt), and the round-3 entry points (partial aggregate, run-edge sparse
masks, MT19937 draws, pipelined host twins, recycled device blocks), each compared with the oracle.  The long runs
(hundreds of cases per seed) are in tests/perf/README.md."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "perf", script)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, OMP_WAIT_POLICY="passive"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_chained_launches(seed):
    assert "FUZZ_OK 120 cases" in _run("fuzz_chains.py", 120, seed)


def test_fuzz_sparse_round():
    assert "FUZZ_SPARSE_OK 20 cases" in _run("fuzz_sparse.py", 20, 103)


def test_fuzz_round3_entry_points():
    assert "FUZZ_R3_OK 64 cases" in _run("fuzz_round3.py", 8, 104)


def test_fuzz_round4_entry_points():
    """The round-4 entry points on random shapes against the oracle: every client's encrypt on an element slice (+ the slice of the sum),
    the fused codec over a flattened model with a random layer table, the ctx-resident precompute with dropouts, the uint32 reduce, the
    span-bounds handle, dynamic_masking's cost on the device, the backward carry walk of the element-sharded packed reduce, the clients'
    sparse encrypts + the aggregate of their uploads in one pass (ragged and clustered lists)."""
    assert "FUZZ_R4_OK 72 cases" in _run("fuzz_round4.py", 8, 105)


def test_fuzz_round5_entry_points():
    """What round 5 added, on random shapes against the oracle / NumPy: the compact layout at its compile-time widths and int_bits 64 at
    compile time in paired-kernel launches (chunk ends everywhere, misaligned and in-place vectors, sub-ranges that cut a block), the
    encrypts + their sum in one launch and every fall-back of that entry point, the online encrypts + their sum with precomputed masks,
    the rewritten sparsifier passes (random layer tables, ties at the threshold, residuals; one layer and whole models), the idx + 1 range
    check."""
    assert "FUZZ_R5_OK 18 cases" in _run("fuzz_round5.py", 3, 106)
