"""GPU parity tests of the chained PRF launch (prf_chain_kernel): consecutive clients share their PRF streams
(client c's minus stream is client c + 1's add stream, jzf_flashe.py:349-353), so a batch of C encrypts costs
C + 1 AES streams.  Every ciphertext is compared bit for bit with the oracle, which computes each client on its own."""
import os

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


@pytest.mark.parametrize("b,n,J,idx", [
    (64, 100_003, 16, list(range(10))),                    # one block per lane, chain of ten
    (64, 2_300_017, 16, list(range(7, 12))),               # two blocks per lane (software pipelined pairs)
    (23, 61_706, 16, list(range(100))),                    # BASELINE config 3 at the reference-style int_bits: chain of 100, cut for parallelism
    (20, 7_000_001, 7, [3, 4, 5]),                         # m = 6, ragged chunks, pair mode
    (8, 300_000, 16, [5, 6, 7, 20, 21, 9]),                # m = 16; runs broken by non-consecutive prefixes
    (33, 4099, 3, list(range(130))),                       # more outputs than one launch table holds
    (1, 5000, 5, [0, 1]), (64, 1, 1, [0, 1, 2]), (7, 41, 16, [2 ** 32 - 3, 2 ** 32 - 2]),      # n < n_jobs: empty chunks
])
def test_small_chain_batch_vs_oracle(E, oracle, b, n, J, idx):
    """int_bits <= 64 (m = 128 // b elements per AES block, chunk-dependent counters): the chained launch against the oracle's
    per-client encrypts, double and single mask."""
    eng = E.Engine(KEY, b, device=0)
    rng = np.random.Generator(np.random.PCG64(n + b))
    pts = [rng.integers(0, 2 ** min(b, 63), n, dtype=np.uint64) for _ in idx]
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in idx]
    for scheme, name in ((E.SCHEME_DOUBLE, "double"), (E.SCHEME_SINGLE, "single")):
        eng.encrypt_batch_dev(13, idx, scheme, n, J, dpt, 1, dct)
        step = 1 if len(idx) <= 12 else 9
        for v in list(range(0, len(idx), step)) + [len(idx) - 1]:
            got = dct[v].download(np.uint64, n).reshape(n, 1)
            assert np.array_equal(got, oracle.encrypt(KEY, 13, idx[v], name, J, b, pts[v])), (b, n, name, v)


def test_small_chain_ranges_and_mask_precompute(E, oracle):
    """Chained job lists over ragged element ranges of a chunked vector (block-misaligned first / last elements), with and
    without input, b = 23 and b = 64."""
    for b, n, J in [(23, 61_706, 16), (64, 50_001, 8)]:
        eng = E.Engine(KEY, b, device=0)
        rng = np.random.Generator(np.random.PCG64(b))
        pt = rng.integers(0, 2 ** 20, n, dtype=np.uint64)
        dpt = eng.upload(pt)
        masks = {i: oracle.mask(KEY, 9, i, n, J, b) for i in range(6)}
        first, count = 1237, n - 3001
        outs = [eng.alloc_vec(n) for _ in range(6)]
        jobs = [(c, c + 1, first, count, dpt.ptr + 8 * first, 1, outs[c]) for c in range(3)]         # a chain 0-1-2-3 on a ragged range
        jobs += [(c, c + 1, 0, n, None, 0, outs[c]) for c in range(3, 5)]                            # mask precompute chain 3-4-5
        jobs.append((5, 0, 17, 1000, dpt.ptr + 8 * 17, 1, outs[5]))                                  # an unrelated job
        eng.prf_jobs_dev(9, n, J, jobs)
        ptl = pt.reshape(n, 1)
        z = np.zeros((n, 1), dtype=np.uint64)
        for c in range(3):
            want = oracle.combine(b, ptl[first:first + count], masks[c][first:first + count], masks[c + 1][first:first + count])
            assert np.array_equal(outs[c].download(np.uint64, count).reshape(count, 1), want), (b, c)
        for c in range(3, 5):
            assert np.array_equal(outs[c].download(np.uint64, n).reshape(n, 1), oracle.combine(b, z, masks[c], masks[c + 1])), (b, c)
        assert np.array_equal(outs[5].download(np.uint64, 1000).reshape(1000, 1), oracle.combine(b, ptl[17:1017], masks[5][17:1017], masks[0][17:1017])), b


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


@pytest.mark.parametrize("b,n,idx,scheme", [
    (128, 2_300_017, list(range(7, 12)), "double"),        # ONE launch: whole tiles + half-tile tails, ragged end
    (128, 4_200_001, list(range(10)), "double"),           # ONE launch: config-2 shape, shorter
    (100, 2_200_013, [0, 1, 2], "double"),                 # ONE launch, b < 128: the sum is masked like every ciphertext
    (128, 100_003, list(range(10)), "double"),             # too short to fill the chip uncut: encrypts, then the reduce
    (128, 2_300_017, [5, 6, 7, 20, 21], "double"),         # broken run (two chains): encrypts, then the reduce
    (128, 2_200_013, [3, 4, 5], "single"),                 # single mask: encrypts, then the reduce
    (64, 300_007, [0, 1, 2, 3], "double"), (20, 50_001, [2, 3], "double"),      # int_bits <= 64
    (128, 3000, list(range(130)), "double"),               # more outputs than one launch holds
])
def test_encrypt_batch_with_partial_aggregate(E, oracle, b, n, idx, scheme):
    """flashe_encrypt_batch_sum_dev: every ciphertext equals the oracle's encrypt, and the partial aggregate equals the oracle's
    element-wise reduce (jzf_aggregator.py:424-430) of those ciphertexts -- in the one-launch form (running sum in registers) and in
    every shape that falls back to encrypt + reduce."""
    eng = E.Engine(KEY, b, device=0)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(n + len(idx) + b))
    pts = [rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for _ in idx]
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in idx]
    dsum = eng.alloc_vec(n)
    eng._check(eng._lib.flashe_memset_dev(eng._h, dsum.ptr, 0xA5, dsum.nbytes))
    eng.encrypt_batch_sum_dev(6, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, 16, dpt, 1, dct, dsum)
    want = [oracle.encrypt(KEY, 6, i, scheme, 16, b, p) for i, p in zip(idx, pts)]
    step = 1 if len(idx) <= 12 else 11
    for v in list(range(0, len(idx), step)) + [len(idx) - 1]:
        assert np.array_equal(dct[v].download(np.uint64, n * Lb).reshape(n, Lb), want[v]), (b, n, v)
    assert np.array_equal(dsum.download(np.uint64, n * Lb).reshape(n, Lb), oracle.aggregate_elem(want, b)), (b, n, "partial aggregate")
    with pytest.raises(E.FlasheError):
        eng.encrypt_batch_sum_dev(6, idx, E.SCHEME_DOUBLE, n, 16, dpt, 1, dct, dct[0])      # the sum must not alias a ciphertext
    with pytest.raises(E.FlasheError):
        eng.encrypt_batch_sum_dev(6, idx, E.SCHEME_DOUBLE, n, 16, dpt, 1, dct, dpt[-1])     # ... nor a plaintext (chunk ends add to the sum in memory)


@pytest.mark.parametrize("b,n,J,C,scheme", [(20, 6_400_007, 16, 3, "double"), (23, 5_300_003, 7, 2, "double"), (16, 8_388_608, 1, 2, "double"),
                                           (20, 6_300_001, 5000, 2, "single"), (16, 4_200_000, 16, 3, "single"), (20, 6_299_999, 16, 2, "double"),
                                           (32, 4_200_011, 16, 2, "double"), (24, 5_300_001, 16, 3, "double"), (8, 16_800_003, 7, 2, "double"),
                                           (32, 4_194_304, 999, 2, "single")])
def test_compact_layout_at_compile_time_widths(E, oracle, b, n, J, C, scheme):
    """int_bits = 16 / 20 / 23 / 24 / 32 (FLASHE_FIXED32_WIDTHS, csrc/kernels.hip; 8: the paired kernel at a run-time width) in the compact
    layout, launches long enough for the paired kernel (>= 1 M AES blocks per stream): the
    instantiations of prf_small_chain_kernel with the width compiled in (a lane loads, adds and stores its own block's 128 // b elements,
    slot positions constant).  EVERY ciphertext word against the oracle's encrypt (jzf_flashe.py:19-45 slot order and chunk-dependent
    counters, :456-488): whole tiles take the new path; chunk ends (n_jobs = 7, 16, 5000: thousands of partial blocks), the ragged end
    of the vector and tiles that straddle a chunk take the general walk inside the same kernel.  Plaintexts use all b bits so that the
    per-slot subtraction wraps; misaligned vectors (4 bytes off a 16-byte boundary) and in-place encryption are covered too."""
    eng = E.Engine(KEY, b, device=0)
    assert eng.compact_supported()
    rng = np.random.Generator(np.random.PCG64(n + 17 * C + b))
    pts = [rng.integers(0, 2 ** b, n, dtype=np.uint64) for _ in range(C)]
    idx = list(range(2, 2 + C))
    sch = E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE
    d32 = [eng.upload(p.astype(np.uint32)) for p in pts]
    c32 = [eng.alloc(4 * n + 32) for _ in range(C)]
    eng.encrypt_batch_u32_dev(4, idx, sch, n, J, d32, c32)
    want = [oracle.encrypt(KEY, 4, i, scheme, J, b, p)[:, 0].astype(np.uint32) for i, p in zip(idx, pts)]
    for v in range(C):
        got = c32[v].download(np.uint32, n)
        bad = np.flatnonzero(got != want[v])
        assert bad.size == 0, (b, n, J, v, bad[:8], got[bad[:4]], want[v][bad[:4]])
    # vectors that start 4 bytes past a 16-byte boundary, output in place
    off = eng.alloc(4 * n + 64)
    off.upload_at(4, pts[0].astype(np.uint32))
    eng.encrypt_batch_u32_dev(4, [idx[0]], sch, n, J, [off.ptr + 4], [off.ptr + 4])
    assert np.array_equal(off.download(np.uint32, n + 1)[1:], want[0]), (b, n, "misaligned, in place")
    # the round: reduce fused with the decrypt of the result gives back the plaintext sum
    if scheme == "double":
        out = eng.alloc(4 * n + 16)
        eng.aggregate_decrypt_u32_dev(4, [idx[-1] + 1], [idx[0]], n, J, 0, n, c32, None, out, 4)
        assert np.array_equal(out.download(np.uint32, n).astype(np.uint64), sum(pts) & np.uint64((1 << b) - 1)), (b, n, "round trip")
    # the encrypts AND the sum of their ciphertexts from one launch (flashe_encrypt_batch_sum_u32_dev): the same ciphertexts, the sum the
    # oracle's element-wise reduce gives (jzf_aggregator.py:424-430) -- blocks of whole tiles keep it in registers, chunk ends in memory;
    # the single mask and short vectors take the two-launch form behind the same entry point
    c2 = [eng.alloc(4 * n + 32) for _ in range(C)]
    dsum = eng.alloc(4 * n + 32)
    eng._check(eng._lib.flashe_memset_dev(eng._h, dsum.ptr, 0x77, dsum.nbytes))
    eng.encrypt_batch_sum_u32_dev(4, idx, sch, n, J, d32, c2, dsum)
    for v in range(C):
        assert np.array_equal(c2[v].download(np.uint32, n), want[v]), (b, n, v, "with the sum")
    wsum = np.zeros(n, dtype=np.uint64)
    for w in want:
        wsum += w
    bad = np.flatnonzero(dsum.download(np.uint32, n).astype(np.uint64) != (wsum & np.uint64((1 << b) - 1)))
    assert bad.size == 0, (b, n, J, "sum of the ciphertexts", bad[:8])
    # and decrypting that one vector is the round
    if scheme == "double":
        out2 = eng.alloc(4 * n + 16)
        eng.aggregate_decrypt_u32_dev(4, [idx[-1] + 1], [idx[0]], n, J, 0, n, [dsum], None, out2, 4)
        assert np.array_equal(out2.download(np.uint32, n).astype(np.uint64), sum(pts) & np.uint64((1 << b) - 1)), (b, n, "round trip through the sum")
    with pytest.raises(E.FlasheError):
        eng.encrypt_batch_sum_u32_dev(4, idx, sch, n, J, d32, c2, c2[0])


@pytest.mark.parametrize("b,n,J,C", [(20, 50_001, 16, 4), (25, 900_000, 16, 3), (8, 4099, 3, 5), (20, 6_400_000, 16, 130)])
def test_compact_encrypt_batch_sum_fallbacks(E, oracle, b, n, J, C):
    """flashe_encrypt_batch_sum_u32_dev on shapes its one-launch form does not carry (short vectors, other widths, more clients than a
    launch holds, clients that are not consecutive): the same ciphertexts and the same sum through the encrypts + the reduce."""
    eng = E.Engine(KEY, b, device=0)
    rng = np.random.Generator(np.random.PCG64(n + b))
    idx = list(range(1, 1 + C)) if C != 4 else [1, 2, 4, 5]
    small = C > 64
    pts = [rng.integers(0, 2 ** b, n, dtype=np.uint64) for _ in range(2 if small else C)]
    d32 = [eng.upload(pts[v % len(pts)].astype(np.uint32)) for v in range(C)]
    c32 = [eng.alloc(4 * n + 16) for _ in range(C)]
    dsum = eng.alloc(4 * n + 16)
    eng.encrypt_batch_sum_u32_dev(2, idx, E.SCHEME_DOUBLE, n, J, d32, c32, dsum)
    wsum = np.zeros(n, dtype=np.uint64)
    for v in range(C):
        w = oracle.encrypt(KEY, 2, idx[v], "double", J, b, pts[v % len(pts)])[:, 0]
        wsum += w
        if v in (0, C // 2, C - 1):
            assert np.array_equal(c32[v].download(np.uint32, n).astype(np.uint64), w), (b, n, v)
    assert np.array_equal(dsum.download(np.uint32, n).astype(np.uint64), wsum & np.uint64((1 << b) - 1)), (b, n, C)


@pytest.mark.parametrize("n,J,C,scheme", [(2_400_001, 16, 3, "double"), (2_300_000, 7, 2, "single"), (2_200_003, 3001, 2, "double")])
def test_int_bits_64_at_compile_time(E, oracle, n, J, C, scheme):
    """int_bits = 64 in launches long enough for the paired kernel: the instantiation with the width compiled in (the two elements of a
    block as one 16-byte access, plaintext requested before the AES rounds).  Every ciphertext word against the oracle (jzf_flashe.py:19-45:
    low half first, chunk-dependent counters); chunk ends (an odd chunk length leaves a one-element block), the ragged end of the vector
    and a sub-range whose first element is the SECOND half of a block take the per-element form inside the same kernel."""
    eng = E.Engine(KEY, 64, device=0)
    rng = np.random.Generator(np.random.PCG64(n + C))
    pts = [rng.integers(0, 2 ** 64, n, dtype=np.uint64) for _ in range(C)]
    idx = list(range(4, 4 + C))
    sch = E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in range(C)]
    eng.encrypt_batch_dev(8, idx, sch, n, J, dpt, 1, dct)
    want = [oracle.encrypt(KEY, 8, i, scheme, J, 64, p)[:, 0] for i, p in zip(idx, pts)]
    for v in range(C):
        got = dct[v].download(np.uint64, n)
        bad = np.flatnonzero(got != want[v])
        assert bad.size == 0, (n, J, v, bad[:8])
    # a sub-range that starts one element into the vector (8 bytes off a 16-byte boundary, mid-block)
    first, count = 1, n - 2
    out = eng.alloc_vec(n)
    eng.encrypt_batch_range_dev(8, idx[:2], sch, n, J, first, count, [d.ptr + 8 * first for d in dpt[:2]], 1, [out.ptr + 8 * first, dct[0].ptr + 8 * first])
    assert np.array_equal(out.download(np.uint64, n)[first:first + count], want[0][first:first + count])


@pytest.mark.parametrize("b,n,J,C,scheme", [(20, 100_003, 16, 10, "double"), (32, 70_001, 3, 4, "double"), (23, 61_706, 16, 100, "double"),
                                           (16, 2_000_003, 16, 3, "double"), (8, 4099, 1, 2, "single"), (1, 777, 5, 2, "double"),
                                           (31, 12, 16, 3, "double"), (20, 1_500_000, 16, 129, "single"), (25, 300_000, 7, 12, "double")])
def test_compact_u32_layout_equals_the_one_limb_layout(E, oracle, b, n, J, C, scheme):
    """flashe_encrypt_batch_u32_dev / flashe_aggregate_decrypt_u32_dev / widen / narrow (int_bits <= 32, the same values as uint32
    arrays): every ciphertext equals the oracle's encrypt (jzf_flashe.py:456-488) truncated to 32 bits, the fused reduce + decrypt
    equals aggregate (jzf_aggregator.py:424-430) + decrypt (jzf_flashe.py:570-571) on full and odd sub-ranges, with uint32 and uint64
    results, with and without the stored aggregate; wider moduli and prefix lists are refused."""
    eng = E.Engine(KEY, b, device=0)
    rng = np.random.Generator(np.random.PCG64(n + C + b))
    pts = [rng.integers(0, 2 ** b, n, dtype=np.uint64) for _ in range(C)]
    idx = list(range(5, 5 + C))
    d32 = [eng.upload(p.astype(np.uint32)) for p in pts]
    c32 = [eng.alloc(4 * n + 16) for _ in range(C)]
    eng.encrypt_batch_u32_dev(9, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, J, d32, c32)
    want = [oracle.encrypt(KEY, 9, i, scheme, J, b, p) for i, p in zip(idx, pts)]
    for v in sorted({0, C - 1, C // 2}):
        assert np.array_equal(c32[v].download(np.uint32, n), want[v][:, 0].astype(np.uint32)), (b, n, v)
    # widen: the compact ciphertext as a one-limb vector; narrow: back
    w = eng.alloc_vec(n)
    eng.widen_u32_dev(n, c32[0], w)
    assert np.array_equal(w.download(np.uint64, n), want[0][:, 0])
    back = eng.alloc(4 * n + 16)
    eng.narrow_u32_dev(n, w, back)
    assert np.array_equal(back.download(np.uint32, n), want[0][:, 0].astype(np.uint32))
    if C > 64:
        with pytest.raises(E.FlasheError):
            eng.aggregate_decrypt_u32_dev(9, [3], [4], n, J, 0, n, c32, None, w)
        return
    agg = oracle.aggregate_elem(want, b)
    for add, minus in (([idx[-1] + 1], [idx[0]]), ([7], [])):
        full = oracle.combine(b, agg, oracle.mask_sum(KEY, 9, add, n, J, b), oracle.mask_sum(KEY, 9, minus, n, J, b))
        for first, count in ((0, n), (1, n - 1), (min(n - 1, 257), max(1, (n - 257) // 2)), (n - 1, 1)):
            if first + count > n:
                continue
            ptrs = [c.ptr + 4 * first for c in c32]
            for out_bytes in (8, 4):
                out, ao = eng.alloc(8 * count + 16), eng.alloc(8 * count + 16)
                eng.aggregate_decrypt_u32_dev(9, add, minus, n, J, first, count, ptrs, ao, out, out_bytes)
                dt = np.uint64 if out_bytes == 8 else np.uint32
                assert np.array_equal(out.download(dt, count).astype(np.uint64), full[first:first + count, 0]), (b, n, first, count, out_bytes, add)
                assert np.array_equal(ao.download(dt, count).astype(np.uint64), agg[first:first + count, 0]), (b, n, first, count, out_bytes)
    with pytest.raises(E.FlasheError):
        eng.aggregate_decrypt_u32_dev(9, [3, 4], [0], n, J, 0, n, c32, None, w)          # prefix lists: widen and use the general call
    wide = E.Engine(KEY, 40, device=0)
    with pytest.raises(E.FlasheError):
        wide.encrypt_batch_u32_dev(9, idx, E.SCHEME_DOUBLE, n, J, d32, c32)


_PROBE_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from flashe_amd.engine import Engine
from oracle import flashe_oracle as orc
orc.build()
KEY = bytes(range(32))
# b = 128: ten chained encrypts of a vector long enough for whole tiles, and the reduce fused with the decrypt
n, C, it = 300_000, 4, 3
eng = Engine(KEY, 128, device=0)
rng = np.random.Generator(np.random.PCG64(5))
pts = [rng.integers(0, 2 ** 64, n, dtype=np.uint64) for _ in range(C)]
dp = [eng.upload(p) for p in pts]
dc = [eng.alloc_vec(n) for _ in range(C)]
eng.encrypt_batch_dev(it, list(range(C)), 1, n, 16, dp, 1, dc)
cts = [d.download(np.uint64, 2 * n).reshape(n, 2) for d in dc]
for c in range(C):
    assert np.array_equal(cts[c], orc.encrypt(KEY, it, c, "double", 16, 128, pts[c])), ("encrypt", c)
out = eng.alloc_vec(n)
eng.aggregate_decrypt_range_dev(it, [C], [0], n, 16, 0, n, dc, None, out)
want = orc.decrypt(KEY, it, [C], [0], 16, 128, orc.aggregate_elem(cts, 128))
assert np.array_equal(out.download(np.uint64, 2 * n).reshape(n, 2), want), "reduce + decrypt"
# b = 20: the small-modulus reduce fused with the decrypt (the launch FLASHE_SMALL_REDUCE_PROBE used to strip of its AES rounds)
e20 = Engine(KEY, 20, device=0)
p20 = [rng.integers(0, 2 ** 16, n, dtype=np.uint64) for _ in range(C)]
d20 = [e20.upload(p) for p in p20]
c20 = [e20.alloc_vec(n) for _ in range(C)]
e20.encrypt_batch_dev(it, list(range(C)), 1, n, 7, d20, 1, c20)
h20 = [d.download(np.uint64, n).reshape(n, 1) for d in c20]
for c in range(C):
    assert np.array_equal(h20[c], orc.encrypt(KEY, it, c, "double", 7, 20, p20[c])), ("encrypt b=20", c)
o20 = e20.alloc_vec(n)
e20.aggregate_decrypt_range_dev(it, [C], [0], n, 7, 0, n, c20, None, o20)
w20 = orc.decrypt(KEY, it, [C], [0], 7, 20, orc.aggregate_elem(h20, 20))
assert np.array_equal(o20.download(np.uint64, n).reshape(n, 1), w20), "reduce + decrypt b=20"
assert np.array_equal(w20[:, 0], sum(p20) & np.uint64((1 << 20) - 1))
print("PROBES-IGNORED-OK")
"""


def test_probe_variables_cannot_make_the_product_library_skip_work():
    """FLASHE_CHAIN_TUNE + FLASHE_CHAIN_PROBE made the encrypt launch return after its table fill (rc 0, no ciphertext
    written) and FLASHE_SMALL_REDUCE_PROBE made the b <= 64 reduce + decrypt skip its AES rounds.  Those branches are compiled out of
    libflashe_hip.so: a process started with every probe / tuning variable set gets oracle-equal ciphertexts and results."""
    import subprocess
    import sys
    env = dict(os.environ, FLASHE_CHAIN_TUNE="1", FLASHE_CHAIN_PROBE="1", FLASHE_CHAIN_HALF="1", FLASHE_CHAIN_PARTS="3", FLASHE_CHAIN_GRID="7",
               FLASHE_SMALL_REDUCE_PROBE="1", FLASHE_SMALL_REDUCE_CB="8", FLASHE_SMALL_REDUCE_SPLIT="0", FLASHE_HYBRID_BS_PERMILLE="500",
               FLASHE_BS16_WAVES="2", FLASHE_SMALL_DIRECT="0", FLASHE_SMALL_LATENCY="0", FLASHE_SMALL_FUSED_REDUCE="0", FLASHE_MT_PARALLEL="0")
    env.pop("FLASHE_LIB_NAME", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _PROBE_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PROBES-IGNORED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
