"""GPU parity tests of the chained PRF launch (prf_chain_kernel): consecutive clients share their PRF streams
(client c's minus stream is client c + 1's add stream, jzf_flashe.py:349-353), so a batch of C encrypts costs
C + 1 AES streams.  Every ciphertext is compared bit for bit with the oracle, which computes each client on its own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))


@pytest.fixture(scope="module")
def E():
    from flashe_amd import engine
    return engine


def L(b):
    return 2 if b > 64 else 1


@pytest.mark.parametrize("b,n,idx", [
    (128, 100_003, list(range(10))),                       # half-tile mode, one chain of ten
    (128, 2_300_017, list(range(7, 12))),                  # whole tiles + half-tile tails, ragged end
    (128, 61_706, list(range(100))),                       # BASELINE config 3 shape: a chain of 100, cut for parallelism
    (128, 3000, list(range(130))),                         # more outputs than one launch table holds
    (100, 70_001, [5, 6, 7, 20, 21, 9, 2 ** 32 - 2]),      # runs broken by non-consecutive prefixes; idx + 1 = 2^32 - 1
    (65, 1, [0, 1]), (128, 255, [3, 4, 5]), (128, 257, [3, 4, 5]), (127, 4099, [0]),
])
def test_chain_batch_vs_oracle(E, oracle, b, n, idx):
    eng = E.Engine(KEY, b, device=0)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(n + len(idx)))
    pts = [rng.integers(0, 2 ** 64, n, dtype=np.uint64) for _ in idx]
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in idx]
    for scheme, name in ((E.SCHEME_DOUBLE, "double"), (E.SCHEME_SINGLE, "single")):
        eng.encrypt_batch_dev(13, idx, scheme, n, 16, dpt, 1, dct)
        step = 1 if len(idx) <= 12 else 7
        for v in list(range(0, len(idx), step)) + [len(idx) - 1]:
            got = dct[v].download(np.uint64, n * Lb).reshape(n, Lb)
            assert np.array_equal(got, oracle.encrypt(KEY, 13, idx[v], name, 16, b, pts[v])), (b, n, name, v)


def test_chain_two_limb_inputs_and_no_input(E, oracle):
    """16-byte plaintext containers (in_limbs = 2) and bare mask differences (in = NULL) through chained job lists."""
    b, n, it = 128, 70_001, 4
    eng = E.Engine(KEY, b, device=0)
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.integers(0, 2 ** 64, size=(n, 2), dtype=np.uint64)
    dx = eng.upload(x)
    masks = {i: oracle.mask(KEY, it, i, n, 16, b) for i in range(6)}
    # a chain 0-1-2-3 over a ragged range, then an unrelated job, then a chain 4-5 without input
    first, count = 777, n - 1500
    outs = [eng.alloc_vec(n) for _ in range(5)]
    jobs = [(c, c + 1, first, count, dx.ptr + 16 * first, 2, outs[c]) for c in range(3)]
    jobs.append((5, 0, 0, n, dx, 2, outs[3]))
    jobs.append((4, 5, 256, 1024, None, 0, outs[4]))
    eng.prf_jobs_dev(it, n, 16, jobs)
    z = np.zeros((n, 2), dtype=np.uint64)
    for c in range(3):
        want = oracle.combine(b, x[first:first + count], masks[c][first:first + count], masks[c + 1][first:first + count])
        assert np.array_equal(outs[c].download(np.uint64, 2 * count).reshape(count, 2), want), c
    assert np.array_equal(outs[3].download(np.uint64, 2 * n).reshape(n, 2), oracle.combine(b, x, masks[5], masks[0]))
    assert np.array_equal(outs[4].download(np.uint64, 2 * 1024).reshape(1024, 2), oracle.combine(b, z[:1024], masks[4][256:1280], masks[5][256:1280]))


def test_chain_matches_unchained_path(E, oracle, monkeypatch):
    """FLASHE_CHAIN=0 (every job computes both of its streams) and the chained launch write identical ciphertexts."""
    b, n, C = 128, 1_200_003, 6
    rng = np.random.Generator(np.random.PCG64(99))
    pts = [rng.integers(0, 2 ** 64, n, dtype=np.uint64) for _ in range(C)]
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FLASHE_CHAIN", flag)
        eng = E.Engine(KEY, b, device=0)
        dpt = [eng.upload(p) for p in pts]
        dct = [eng.alloc_vec(n) for _ in range(C)]
        eng.encrypt_batch_dev(2, list(range(40, 40 + C)), E.SCHEME_DOUBLE, n, 16, dpt, 1, dct)
        res.append([d.download(np.uint64, 2 * n) for d in dct])
    for c in range(C):
        assert np.array_equal(res[0][c], res[1][c]), c
    assert np.array_equal(res[0][2].reshape(n, 2), oracle.encrypt(KEY, 2, 42, "double", 16, b, pts[2]))


def test_chain_counter_window_fallback(E):
    """A chained job list whose range straddles a 2^32 counter boundary cannot take the CTR shortcuts: the call falls back to
    the job-table kernel's generic path; one above 2^32 stays on the chained path (high counter word folded into the prefix
    words).  Expected values from the host AES, element by element."""
    b, it = 128, 1
    eng = E.Engine(KEY, b, device=0)
    n = 2 ** 33
    for first, count in [(2 ** 32 - 3000, 6000), (2 ** 32 + 256 * 5 + 17, 70_000)]:
        rng = np.random.Generator(np.random.PCG64(count))
        pt = rng.integers(0, 2 ** 64, count, dtype=np.uint64)
        dpt = eng.upload(pt)
        outs = [eng.alloc_vec(count) for _ in range(3)]
        eng.prf_jobs_dev(it, n, 1, [(c, c + 1, first, count, dpt, 1, outs[c]) for c in range(3)])
        for c in range(3):
            got = outs[c].download(np.uint64, 2 * count).reshape(count, 2)
            for e in list(range(0, count, 997)) + [2999, 3000, 3001, count - 1]:
                if e >= count:
                    continue
                ctr = first + e
                blk = lambda i: int.from_bytes(E.prp_block(KEY, it.to_bytes(4, "big") + i.to_bytes(4, "big") + ctr.to_bytes(8, "big")), "big")
                want = (int(pt[e]) + blk(c) - blk(c + 1)) % (1 << 128)
                assert int(got[e, 0]) | (int(got[e, 1]) << 64) == want, (first, c, e)
