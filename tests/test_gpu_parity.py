"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of
libflashe_hip.so and is compared bit-for-bit with (a) the golden vectors generated from the
reference and (b) the CPU oracle on the same seeded inputs; full BASELINE sizes are covered by
size-independent properties (round trip, linearity) plus one full-array oracle comparison."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, unhex

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))
# The bit-sliced PRF backends live in a separate library (make -C flashe_amd/csrc bitslice): tests/test_gpu_bitslice.py re-runs the
# backend-parametrised tests of this file in a process that loads it (FLASHE_LIB_NAME) with this switch on.
BITSLICE_BACKENDS = [2, 4] if os.environ.get("FLASHE_TEST_BITSLICE") == "1" else []


@pytest.fixture(scope="module")
def E():
    from flashe_amd import engine
    return engine


def make(E, b, key=KEY):
    return E.Engine(key, b, device=0)


def L(b):
    return 2 if b > 64 else 1


def rand_limbs(rng, n, b):
    out = rng.integers(0, 2 ** 64, size=(n, L(b)), dtype=np.uint64)
    if b > 64 and b < 128:
        out[:, 1] &= np.uint64((1 << (b - 64)) - 1)
    elif b < 64:
        out[:, 0] &= np.uint64((1 << b) - 1)
    return out


# ------------------------------------------------------------------ known answers
def test_device_selftest(E):
    for b in (128, 64, 20):
        make(E, b).selftest()


def test_mask_streams_golden(E, oracle):
    g = load_golden("mask_streams.json")
    engines = {}
    for c in g["cases"]:
        eng = engines.setdefault(c["b"], make(E, c["b"]))
        got = oracle.limbs_to_ints(eng.mask(c["iter"], [c["idx"]], c["n"], c["n_jobs"]))
        assert got == unhex(c["stream"]), (c["b"], c["n"], c["n_jobs"])
    for c in g["sums"]:
        eng = engines.setdefault(c["b"], make(E, c["b"]))
        assert oracle.limbs_to_ints(eng.mask(c["iter"], c["add_idx"], c["n"], c["n_jobs"])) == unhex(c["add"])
        assert oracle.limbs_to_ints(eng.mask(c["iter"], c["minus_idx"], c["n"], c["n_jobs"])) == unhex(c["minus"])


def test_cipher_rounds_golden_engine(E, oracle):
    for c in load_golden("cipher_rounds.json")["cases"]:
        b, n, J, it = c["b"], c["n"], c["n_jobs"], c["iter"]
        sch = E.SCHEME_DOUBLE if c["scheme"] == "double" else E.SCHEME_SINGLE
        eng = make(E, b)
        cts = {}
        for i, pt in c["pt"].items():
            ct = eng.encrypt(it, int(i), sch, J, oracle.ints_to_limbs(unhex(pt), b))
            assert oracle.limbs_to_ints(ct) == unhex(c["ct"][i])
            cts[int(i)] = ct
        models = [cts[i] for i in c["uploaded"]]
        if n == 0:
            continue
        agg = eng.aggregate_elem(models)
        assert oracle.limbs_to_ints(agg) == unhex(c["agg_elem"])
        packed = [eng.pack(m) for m in models]
        aggp = eng.aggregate_packed(packed, n * b)
        assert sum(int(v) << (64 * i) for i, v in enumerate(aggp)) == int(c["agg_packed_int"], 16)
        aggp_el = eng.unpack(aggp, n)
        assert oracle.limbs_to_ints(aggp_el) == unhex(c["agg_packed"])
        if c["scheme"] == "double":
            add_idx, minus_idx = E.telescope(list(c["uploaded"]))
        else:
            add_idx, minus_idx = [], list(c["uploaded"])
        assert oracle.limbs_to_ints(eng.decrypt(it, add_idx, minus_idx, J, agg)) == unhex(c["dec_elem"])
        assert oracle.limbs_to_ints(eng.decrypt(it, add_idx, minus_idx, J, aggp_el)) == unhex(c["dec_packed"])


def test_flashe_cipher_class_golden(E):
    """The same scenarios the CPU suite runs on the engine double, now on the HIP engine."""
    from flashe_amd import cipher as cm
    from cipher_scenarios import (run_precompute_case, run_round_case, run_sparse_dense_double_case,
                                  run_sparse_single_case)
    assert cm.FlasheCipher._engine_cls is E.Engine
    for c in load_golden("cipher_rounds.json")["cases"]:
        run_round_case(cm, c)
    for c in load_golden("precompute.json")["cases"]:
        run_precompute_case(cm, c)
    g = load_golden("sparse.json")
    for c in g["single"]:
        run_sparse_single_case(cm, c)
    for c in g["dense_double"]:
        run_sparse_dense_double_case(cm, c)


def test_pack_golden(E, oracle):
    g = load_golden("pack.json")
    for c in g["cases"] + g["merges"]:
        eng = make(E, c["b"])
        v = oracle.ints_to_limbs(unhex(c["vals"]), c["b"])
        p = eng.pack(v)
        assert sum(int(x) << (64 * i) for i, x in enumerate(p)) == int(c["packed_int"], 16)
        assert oracle.limbs_to_ints(eng.unpack(p, c["n"])) == unhex(c["vals"])
    ce = g["carry_example"]
    eng = make(E, 8)
    a, c2 = oracle.ints_to_limbs(unhex(ce["a"]), 8), oracle.ints_to_limbs(unhex(ce["c"]), 8)
    s = eng.aggregate_packed([eng.pack(a), eng.pack(c2)], 24)
    assert oracle.limbs_to_ints(eng.unpack(s, 3)) == unhex(ce["packed_sum"])
    assert oracle.limbs_to_ints(eng.aggregate_elem([a, c2])) == unhex(ce["elem_sum"])


def test_sparse_golden_engine(E, oracle):
    g = load_golden("sparse.json")
    for c in g["single"]:
        b, total, J, it, C = c["b"], c["total"], c["n_jobs"], c["iter"], c["num_clients"]
        eng = make(E, b)
        dense = []
        for i in range(C):
            up = unhex(c["uploads"][i])
            ct = eng.encrypt(it, i, E.SCHEME_SINGLE, J, oracle.ints_to_limbs(unhex(c["pt"][i]), b))
            assert oracle.limbs_to_ints(ct) == up[:-1]
            d = eng.expand_to_dense(total, c["locs"][i], ct, oracle.ints_to_limbs([up[-1]], b))
            assert oracle.limbs_to_ints(d) == unhex(c["dense"][i])
            dense.append(d)
        agg = eng.aggregate_elem(dense)
        assert oracle.limbs_to_ints(agg) == unhex(c["agg"])
        mm = eng.sparse_minus_mask(it, c["locs"], total, J)
        assert oracle.limbs_to_ints(mm) == unhex(c["minus_mask"])
        assert oracle.limbs_to_ints(eng.combine(agg, None, mm)) == unhex(c["dec"])
    for c in g["dense_double"]:
        eng = make(E, c["b"])
        assert oracle.limbs_to_ints(eng.sparse_dense_mask(c["iter"], c["add_sel"], c["total"])) == unhex(c["add"])
        assert oracle.limbs_to_ints(eng.sparse_dense_mask(c["iter"], c["minus_sel"], c["total"])) == unhex(c["minus"])


def test_config1_golden(E):
    """BASELINE config 1 (1e4 fp32 -> 32-bit quantise -> 64-bit modulus, 2 clients, single mask)."""
    z = np.load(os.path.join(GOLDEN, "config1.npz"))
    n, b, J = 10000, 64, 8
    eng = make(E, b)
    cts = []
    for c in range(2):
        ct = eng.encrypt(0, c, E.SCHEME_SINGLE, J, z[f"q{c}"])
        assert np.array_equal(ct[:, 0], z[f"ct{c}"])
        cts.append(ct)
    agg = eng.aggregate_elem(cts)
    assert np.array_equal(agg[:, 0], z["agg_elem"])
    aggp = eng.unpack(eng.aggregate_packed([eng.pack(c) for c in cts], n * b), n)
    assert np.array_equal(aggp[:, 0], z["agg_packed"])
    assert np.array_equal(eng.decrypt(0, [], [0, 1], J, agg)[:, 0], z["dec_elem"])
    assert np.array_equal(eng.decrypt(0, [], [0, 1], J, aggp)[:, 0], z["dec_packed"])


# ------------------------------------------------------------------ seeded oracle comparisons
@pytest.mark.parametrize("b,n,J", [(128, 100003, 8), (127, 4099, 3), (100, 70001, 16), (65, 5000, 1),
                                   (64, 100003, 8), (63, 20011, 16), (33, 50021, 5), (32, 65536, 16),
                                   (23, 61706, 16), (20, 262144, 16), (16, 30011, 7), (8, 40009, 16),
                                   (7, 10007, 3), (1, 12345, 16), (128, 1, 16), (20, 5, 16), (64, 1025, 1024)])
def test_encrypt_decrypt_vs_oracle(E, oracle, b, n, J):
    rng = np.random.Generator(np.random.PCG64(b * 1000 + J))
    eng = make(E, b)
    it = 123456
    pt = rand_limbs(rng, n, b)
    for sch, name in ((E.SCHEME_DOUBLE, "double"), (E.SCHEME_SINGLE, "single")):
        ct = eng.encrypt(it, 5, sch, J, pt)
        assert np.array_equal(ct, oracle.encrypt(KEY, it, 5, name, J, b, pt)), (b, n, J, name)
    add_idx, minus_idx = [3, 11, 11, 2 ** 32 - 1], [0, 7, 9]
    ct = rand_limbs(rng, n, b)
    assert np.array_equal(eng.decrypt(it, add_idx, minus_idx, J, ct), oracle.decrypt(KEY, it, add_idx, minus_idx, J, b, ct))
    assert np.array_equal(eng.decrypt(it, [], [4], J, ct), oracle.decrypt(KEY, it, [], [4], J, b, ct))
    assert np.array_equal(eng.decrypt(it, [10], [0], J, ct), oracle.decrypt(KEY, it, [10], [0], J, b, ct))
    assert np.array_equal(eng.decrypt(it, [], [], J, ct), oracle.combine(b, ct))
    assert np.array_equal(eng.mask(it, [9], n, J), oracle.mask(KEY, it, 9, n, J, b))


@pytest.mark.parametrize("backend", BITSLICE_BACKENDS)
@pytest.mark.parametrize("b,n", [(128, 100003), (128, 1024), (128, 1), (128, 2049), (127, 4099), (65, 70001), (100, 1000000)])
def test_bitsliced_prf_backend_vs_oracle(E, oracle, b, n, backend):
    """The bit-sliced VALU PRF kernels (2: 32 blocks per lane, 4: packed, 16 blocks per lane) must give the
    same bits as the LDS-table kernel and the oracle."""
    rng = np.random.Generator(np.random.PCG64(b + n))
    eng = make(E, b)
    eng.set_prf_backend(backend)
    it = 77
    pt = rand_limbs(rng, n, b)
    ct = eng.encrypt(it, 5, E.SCHEME_DOUBLE, 16, pt)
    assert np.array_equal(ct, oracle.encrypt(KEY, it, 5, "double", 16, b, pt))
    ct1 = eng.encrypt(it, 2 ** 32 - 1, E.SCHEME_SINGLE, 16, pt)
    assert np.array_equal(ct1, oracle.encrypt(KEY, it, 2 ** 32 - 1, "single", 16, b, pt))
    u64 = pt[:, 0].copy()
    assert np.array_equal(eng.encrypt(it, 0, E.SCHEME_DOUBLE, 16, u64), oracle.encrypt(KEY, it, 0, "double", 16, b, u64))
    assert np.array_equal(eng.decrypt(it, [10], [0], 16, ct), oracle.decrypt(KEY, it, [10], [0], 16, b, ct))
    assert np.array_equal(eng.mask(it, [9], n, 16), oracle.mask(KEY, it, 9, n, 16, b))
    # range twin: a slice in the middle of the vector, unaligned to the 1024-element tiles
    if n > 3000:
        first, count = 1537, n - 2600
        d_in, d_out = eng.upload(ct[first:first + count]), eng.alloc_vec(count)
        eng.decrypt_range_dev(it, [10], [0], n, 16, first, count, d_in, d_out)
        got = d_out.download(np.uint64, count * 2).reshape(count, 2)
        assert np.array_equal(got, oracle.decrypt(KEY, it, [10], [0], 16, b, ct)[first:first + count])
    eng.set_prf_backend(1)
    assert np.array_equal(eng.encrypt(it, 5, E.SCHEME_DOUBLE, 16, pt), ct)


@pytest.mark.parametrize("backend", [1] + BITSLICE_BACKENDS)
def test_counter_window_across_2_32(E, backend):
    """A launch whose counters straddle 2^32 (and one just below / above it) must take the generic
    first-round path; expected values come from the host AES, element by element."""
    eng = make(E, 128)
    eng.set_prf_backend(backend)
    it, idx, n_total = 9, 41, 2 ** 33
    for first, count in [(2 ** 32 - 700, 1500), (2 ** 32 - 3000, 2048), (2 ** 32, 1100), (3 * 2 ** 31 + 5, 64)]:
        rng = np.random.Generator(np.random.PCG64(first % 1000))
        pt = rng.integers(0, 2 ** 64, count, dtype=np.uint64)
        d_in, d_out = eng.upload(pt), eng.alloc_vec(count)
        eng.encrypt_range_dev(it, idx, E.SCHEME_DOUBLE, n_total, 1, first, count, d_in, 1, d_out)
        got = d_out.download(np.uint64, 2 * count).reshape(count, 2)
        for e in list(range(0, count, 97)) + [count - 1, 699, 700, 701]:
            if e >= count:
                continue
            ctr = first + e
            blk = lambda i: int.from_bytes(E.prp_block(KEY, it.to_bytes(4, "big") + i.to_bytes(4, "big") + ctr.to_bytes(8, "big")), "big")
            want = (int(pt[e]) + blk(idx) - blk(idx + 1)) % (1 << 128)
            assert int(got[e, 0]) | (int(got[e, 1]) << 64) == want, (backend, first, e)
        d_m = eng.alloc_vec(count)
        eng.mask_range_dev(it, [idx], n_total, 1, first, count, d_m)          # single-stream mode
        gm = d_m.download(np.uint64, 2 * count).reshape(count, 2)
        for e in (0, count // 2, count - 1):
            ctr = first + e
            want = int.from_bytes(E.prp_block(KEY, it.to_bytes(4, "big") + idx.to_bytes(4, "big") + ctr.to_bytes(8, "big")), "big")
            assert int(gm[e, 0]) | (int(gm[e, 1]) << 64) == want


def test_default_library_has_no_bitsliced_backends(E):
    """The product library contains what runs: asking it for a bit-sliced backend fails loudly and names the build that has them."""
    if BITSLICE_BACKENDS:
        pytest.skip("running against libflashe_hip_bitslice.so")
    eng = make(E, 128)
    for backend in (2, 3, 4):
        with pytest.raises(E.FlasheError, match="make -C flashe_amd/csrc bitslice"):
            eng.set_prf_backend(backend)
    eng.set_prf_backend(1)
    eng.set_prf_backend(0)


def test_counter_window_large_launches(E):
    """Launches big enough for the 4096-element tiles (wave-uniform rounds 1-2 through the scalar cache): above 2^32
    (non-zero high counter word folded into the prefix), straddling 2^32 (generic path on big tiles), and a start
    that is not a multiple of 256 (one-step shortcut on big tiles).  Expected values from the host AES."""
    eng = make(E, 128)
    it, idx, n_total = 5, 1234, 2 ** 34
    for first, count in [(2 ** 32 + 256 * 977, 1_300_000), (2 ** 32 - 600_000, 1_300_000), (3 * 2 ** 32 + 100, 1_100_000), (0, 1_050_000)]:
        rng = np.random.Generator(np.random.PCG64(count))
        pt = rng.integers(0, 2 ** 64, count, dtype=np.uint64)
        d_in, d_out = eng.upload(pt), eng.alloc_vec(count)
        eng.encrypt_range_dev(it, idx, E.SCHEME_DOUBLE, n_total, 1, first, count, d_in, 1, d_out)
        got = d_out.download(np.uint64, 2 * count).reshape(count, 2)
        picks = set(int(v) for v in rng.integers(0, count, 150)) | {0, 63, 64, 255, 256, 4095, 4096, count - 1, count - 1025, 600_000 - 1, 600_000, 600_001}
        for e in sorted(picks):
            if not 0 <= e < count:
                continue
            ctr = first + e
            blk = lambda i: int.from_bytes(E.prp_block(KEY, it.to_bytes(4, "big") + i.to_bytes(4, "big") + ctr.to_bytes(8, "big")), "big")
            want = (int(pt[e]) + blk(idx) - blk(idx + 1)) % (1 << 128)
            assert int(got[e, 0]) | (int(got[e, 1]) << 64) == want, (first, e)
        # linearity against the single-stream launch of the same range: ct - pt == mask(idx) - mask(idx + 1)
        d_a, d_b = eng.alloc_vec(count), eng.alloc_vec(count)
        eng.mask_range_dev(it, [idx], n_total, 1, first, count, d_a)
        eng.mask_range_dev(it, [idx + 1], n_total, 1, first, count, d_b)
        a = d_a.download(np.uint64, 2 * count).reshape(count, 2)
        bb = d_b.download(np.uint64, 2 * count).reshape(count, 2)
        lo = pt + a[:, 0]
        c1 = (lo < pt).astype(np.uint64)
        hi = a[:, 1] + c1
        lo2 = lo - bb[:, 0]
        br = (lo < bb[:, 0]).astype(np.uint64)
        hi2 = hi - bb[:, 1] - br
        assert np.array_equal(got[:, 0], lo2) and np.array_equal(got[:, 1], hi2), first


def test_range_twins_table_backend(E, oracle):
    for b, n, J in [(128, 50000, 16), (64, 50001, 8), (20, 70001, 16), (7, 9999, 3)]:
        rng = np.random.Generator(np.random.PCG64(n))
        eng = make(E, b)
        eng.set_prf_backend(1)
        Lb = L(b)
        ct = rand_limbs(rng, n, b)
        want = oracle.decrypt(KEY, 3, [4, 9], [0, 5], J, b, ct)
        for first, count in [(0, n), (1, n - 1), (1025, 4097), (n - 5, 5), (n // 2, 1), (777, 0)]:
            d_in, d_out = eng.upload(ct[first:first + count] if count else np.zeros((1, Lb), dtype=np.uint64)), eng.alloc_vec(max(count, 1))
            eng.decrypt_range_dev(3, [4, 9], [0, 5], n, J, first, count, d_in, d_out)
            got = d_out.download(np.uint64, count * Lb).reshape(count, Lb)
            assert np.array_equal(got, want[first:first + count]), (b, first, count)
        pt = rand_limbs(rng, n, b)[:, :1].copy()
        wantc = oracle.encrypt(KEY, 3, 6, "double", J, b, pt)
        first, count = 333, n - 1000
        d_in, d_out = eng.upload(pt[first:first + count]), eng.alloc_vec(count)
        eng.encrypt_range_dev(3, 6, E.SCHEME_DOUBLE, n, J, first, count, d_in, 1, d_out)
        assert np.array_equal(d_out.download(np.uint64, count * Lb).reshape(count, Lb), wantc[first:first + count])
        d_out2 = eng.alloc_vec(count)
        eng.mask_range_dev(3, [6, 8], n, J, first, count, d_out2)
        assert np.array_equal(d_out2.download(np.uint64, count * Lb).reshape(count, Lb),
                              oracle.mask_sum(KEY, 3, [6, 8], n, J, b)[first:first + count])


def test_encrypt_batch_equals_single_calls(E, oracle):
    rng = np.random.Generator(np.random.PCG64(77))
    for b, n, nv, sch, name in [(128, 100003, 10, 1, "double"), (128, 1500, 40, 1, "double"), (100, 5000, 3, 0, "single"), (64, 3001, 4, 1, "double"),
                                 (23, 61706, 33, 1, "double"), (20, 9999, 5, 0, "single"), (7, 1000, 2, 1, "double"), (64, 5, 3, 1, "double")]:
        eng = make(E, b)
        Lb = L(b)
        pts = [rng.integers(0, 2 ** min(60, b), n, dtype=np.uint64) for _ in range(nv)]
        idx = [int(v) for v in rng.integers(0, 2 ** 32 - 2, nv)]
        dpt = [eng.upload(p) for p in pts]
        dct = [eng.alloc_vec(n) for _ in range(nv)]
        eng.encrypt_batch_dev(9, idx, sch, n, 16, dpt, 1, dct)
        for v in range(0, nv, max(1, nv // 5)):
            got = dct[v].download(np.uint64, n * Lb).reshape(n, Lb)
            assert np.array_equal(got, oracle.encrypt(KEY, 9, idx[v], name, 16, b, pts[v])), (b, n, v)


@pytest.mark.parametrize("b,n,J", [(128, 70_001, 16), (100, 5000, 3), (64, 30_000, 16), (20, 9999, 7)])
def test_prf_jobs_vs_oracle(E, oracle, b, n, J):
    """flashe_prf_jobs_dev: encrypt, telescoped decrypt and bare mask-difference entries with ragged ranges in one call."""
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(b + n))
    pt = rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64)
    agg = rand_limbs(rng, n, b)
    dpt, dagg = eng.upload(pt), eng.upload(agg)
    masks = {i: oracle.mask(KEY, 6, i, n, J, b) for i in (0, 3, 4, 9, 10)}

    def want(add, minus, first, count, inp):
        z = np.zeros((count, Lb), dtype=np.uint64)
        a = masks[add][first:first + count]
        m = masks[minus][first:first + count] if minus is not None else z
        return oracle.combine(b, inp[first:first + count] if inp is not None else z, a, m)

    pt_l = np.zeros((n, Lb), dtype=np.uint64)
    pt_l[:, 0] = pt
    for has_minus in (True, False):
        spec = [(3, 4, 0, n, "pt"), (10, 0, 0, n, "agg"), (10, 0, 0, n, None), (9, 10, 1025, n - 2000, "pt"), (3, 4, n - 1, 1, "agg"),
                (3, 4, 17, 0, "pt"), (0, 3, 1, 1023, None)]
        outs, jobs = [], []
        for add, minus, first, count, src in spec:
            minus = minus if has_minus else None
            o = eng.alloc_vec(max(count, 1))
            outs.append(o)
            if src == "pt":
                jobs.append((add, minus, first, count, dpt.ptr + 8 * first, 1, o))
            elif src == "agg":
                jobs.append((add, minus, first, count, dagg.ptr + 8 * Lb * first, Lb, o))
            else:
                jobs.append((add, minus, first, count, None, 0, o))
        eng.prf_jobs_dev(6, n, J, jobs)
        for (add, minus, first, count, src), o in zip(spec, outs):
            if count == 0:
                continue
            got = o.download(np.uint64, count * Lb).reshape(count, Lb)
            ref = want(add, minus if has_minus else None, first, count, {"pt": pt_l, "agg": agg, None: None}[src])
            assert np.array_equal(got, ref), (b, has_minus, add, first, count, src)
    # more entries than one launch holds
    many = [(3, 4, 64 * e, 100 + e, None, 0, eng.alloc_vec(200)) for e in range(40)]
    eng.prf_jobs_dev(6, n, J, many)
    for e, job in enumerate(many):
        assert np.array_equal(job[6].download(np.uint64, (100 + e) * Lb).reshape(-1, Lb), want(3, 4, 64 * e, 100 + e, None)), e
    with pytest.raises(Exception):
        eng.prf_jobs_dev(6, n, J, [(3, 4, 0, 10, None, 0, many[0][6]), (3, None, 0, 10, None, 0, many[1][6])])   # mixed has_minus
    with pytest.raises(Exception):
        eng.prf_jobs_dev(6, n, J, [(3, 4, n - 5, 10, None, 0, many[0][6])])                                        # range past n


@pytest.mark.parametrize("b,n,J,C", [(128, 70_001, 16, 10), (100, 5000, 3, 3), (128, 2_100_000, 1, 2), (128, 300_000, 16, 13), (64, 30_000, 16, 4),
                                     (20, 9999, 7, 70), (64, 1_000_003, 16, 10), (20, 1_000_003, 16, 10), (23, 61_706, 16, 5), (7, 4099, 3, 2),
                                     (1, 777, 1, 3), (33, 50_000, 8, 4), (32, 50_001, 5, 11), (16, 200_000, 16, 64), (40, 99_999, 16, 3),
                                     (25, 1100, 2000, 2), (20, 70_000, 16, 17), (64, 70_001, 16, 9)])
def test_aggregate_decrypt_fused_vs_oracle(E, oracle, b, n, J, C):
    """flashe_aggregate_decrypt_range_dev == aggregate_elem followed by decrypt: equally spaced ciphertexts (one launch for
    b > 64), scattered ones, prefix lists, sub-ranges; with and without storing the aggregate.  b <= 64 with one add and at most one
    minus prefix is one launch too (small_reduce_decrypt_kernel: chunk ends inside a wave's tile, more jobs than elements, 64 operands;
    70 operands and prefix lists take the two-launch form)."""
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(n + C))
    cts = [rand_limbs(rng, n, b) for _ in range(C)]
    agg = oracle.aggregate_elem(cts, b)
    n_pad = n + (n & 1)                                                        # keeps every operand 16-byte aligned
    slab = eng.upload(np.concatenate([np.concatenate([c.reshape(-1), np.zeros((n_pad - n) * Lb, dtype=np.uint64)]) for c in cts]))
    scattered = [eng.upload(c) for c in cts]                                   # separate allocations
    cases = [([C], [0]), ([3, 7], [0, 5]), ([], list(range(min(C, 6)))), ([4], [])]
    for add, minus in cases:
        # operands of the reduce: 16-byte aligned for 2-limb vectors = any element; 1-limb vectors need 8 bytes only (a sub-range that
        # starts at an odd element takes the reduce's 8-byte form when the call is not one launch anyway)
        for first, count in ((0, n), (256, n - 777), (1, 1), (n - 1, 1), (78, 0), (77, 500)):
            want = oracle.combine(b, agg[first:first + count], oracle.mask_sum(KEY, 2, add, n, J, b)[first:first + count],
                                  oracle.mask_sum(KEY, 2, minus, n, J, b)[first:first + count]) if count else None
            for layout in ("slab", "scattered"):
                for keep in (True, False):
                    ptrs = [(slab.ptr + 8 * Lb * (c * n_pad + first)) if layout == "slab" else (scattered[c].ptr + 8 * Lb * first) for c in range(C)]
                    out, ao = eng.alloc_vec(max(count, 1)), eng.alloc_vec(max(count, 1))
                    eng.aggregate_decrypt_range_dev(2, add, minus, n, J, first, count, ptrs, ao if keep else None, out)
                    if count == 0:
                        continue
                    assert np.array_equal(out.download(np.uint64, count * Lb).reshape(count, Lb), want), (b, add, minus, first, count, layout, keep)
                    if keep:
                        assert np.array_equal(ao.download(np.uint64, count * Lb).reshape(count, Lb), agg[first:first + count]), (b, layout)
    # the job-list form with a summed input
    if b > 64:
        out, ao = eng.alloc_vec(n), eng.alloc_vec(n)
        eng.prf_jobs_dev(2, n, J, [(C, 0, 0, n, slab, Lb, out, C, n_pad, ao)])
        want = oracle.combine(b, agg, oracle.mask(KEY, 2, C, n, J, b), oracle.mask(KEY, 2, 0, n, J, b))
        assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), want)
        assert np.array_equal(ao.download(np.uint64, n * Lb).reshape(n, Lb), agg)


@pytest.mark.parametrize("b,n,C", [(128, 61_706, 12), (23, 61_706, 6)])
def test_graph_capture_replays_a_round(E, oracle, b, n, C):
    """flashe_graph_*: a whole round (batched encrypt, reduce, decrypt) captured once and replayed on NEW data in the same
    buffers; growth of ctx scratch during a capture is refused."""
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(b))
    dpt = [eng.alloc(8 * n) for _ in range(C)]
    dct = [eng.alloc_vec(n) for _ in range(C)]
    dagg, dout = eng.alloc_vec(n), eng.alloc_vec(n)

    def round_calls():
        eng.encrypt_batch_dev(7, list(range(C)), E.SCHEME_DOUBLE, n, 16, dpt, 1, dct)
        eng.aggregate_elem_dev(dct, n, dagg)
        eng.decrypt_dev(7, [C], [0], n, 16, dagg, dout)

    round_calls()                                   # sizes every scratch buffer
    eng.graph_begin()
    round_calls()
    g = eng.graph_end()
    for rep in range(3):
        pts = [rng.integers(0, 2 ** min(b - 8, 60), n, dtype=np.uint64) for _ in range(C)]
        for d, p in zip(dpt, pts):
            d.upload(p)
        # round `rep` after the captured one: the replay runs with iter 7 + rep (kernel arguments are frozen into the graph, the
        # iter shift is read from device memory), so no round reuses another round's mask streams
        g.launch(iter_shift=rep)
        got = dout.download(np.uint64, n * Lb).reshape(n, Lb)
        want = np.zeros(n, dtype=np.uint64)
        for p in pts:
            want += p
        assert np.array_equal(got[:, 0], want & np.uint64((1 << min(b, 64)) - 1 if b < 64 else 2 ** 64 - 1)), rep
        ct3 = dct[3].download(np.uint64, n * Lb).reshape(n, Lb)
        assert np.array_equal(ct3, oracle.encrypt(KEY, 7 + rep, 3, "double", 16, b, pts[3])), rep
    # the shift is reset after a replay: ordinary calls use the iter they are given
    eng.encrypt_dev(7, 3, E.SCHEME_DOUBLE, n, 16, dpt[3], 1, dct[3])
    assert np.array_equal(dct[3].download(np.uint64, n * Lb).reshape(n, Lb), oracle.encrypt(KEY, 7, 3, "double", 16, b, pts[3]))
    # a graph carries the key schedule it was captured with: after a key change it refuses to replay
    eng.set_key(bytes(range(1, 33)))
    with pytest.raises(E.FlasheError):
        g.launch()
    eng.set_key(KEY)
    # scratch growth inside a capture is refused, the capture still ends cleanly
    big = make(E, 64)
    ops = [big.upload(rng.integers(0, 2 ** 64, 50_000, dtype=np.uint64)) for _ in range(3)]
    o = big.alloc(8 * 50_000)
    big.graph_begin()
    with pytest.raises(Exception):
        big.aggregate_packed_dev(ops, 50_000, 64 * 50_000, o)          # needs the block-summary scratch
    with pytest.raises(Exception):
        big.graph_begin()                                               # already capturing
    try:
        big.graph_end()
    except Exception:
        pass
    big.aggregate_packed_dev(ops, 50_000, 64 * 50_000, o)              # works outside a capture
    big.sync()


def test_prf_jobs_randomised(E, oracle):
    """Seeded sweep of flashe_prf_jobs_dev over bit widths, vector lengths (incl. > 2^20 elements: the 4096-element tiles and
    their scalar-cache shortcut), chunkings and ragged job lists (aligned and unaligned ranges, with and without input)."""
    rng = np.random.Generator(np.random.PCG64(20260))
    cases = [(128, 1_300_000, 16), (128, 300_001, 1), (100, 70_000, 5), (64, 200_000, 16), (33, 10_000, 3), (23, 61_706, 16), (20, 150_001, 7),
             (8, 100_000, 16), (3, 5000, 2), (64, 63, 16), (20, 7, 3)]
    for b, n, J in cases:
        eng = make(E, b)
        Lb = L(b)
        pt = rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64)
        dpt = eng.upload(pt)
        pt_l = np.zeros((n, Lb), dtype=np.uint64)
        pt_l[:, 0] = pt
        idxs = [int(v) for v in rng.integers(0, 2 ** 32 - 1, 3)]
        masks = {i: oracle.mask(KEY, 11, i, n, J, b) for i in idxs}
        for has_minus in (True, False):
            spec, jobs, outs = [], [], []
            for _ in range(int(rng.integers(1, 9))):
                kind = int(rng.integers(0, 4))
                if kind == 0:
                    first, count = 0, n
                elif kind == 1:
                    first = int(rng.integers(0, n))
                    count = int(rng.integers(0, n - first + 1))
                elif kind == 2:
                    first = min(n, int(rng.integers(0, max(1, n // 256 + 1))) * 256)
                    count = n - first
                else:
                    first, count = int(rng.integers(0, n)), 1
                add, minus = idxs[int(rng.integers(0, 3))], idxs[int(rng.integers(0, 3))]
                with_in = bool(rng.integers(0, 2))
                o = eng.alloc_vec(max(count, 1))
                outs.append(o)
                spec.append((add, minus, first, count, with_in))
                jobs.append((add, minus if has_minus else None, first, count, dpt.ptr + 8 * first if with_in else None, 1, o))
            eng.prf_jobs_dev(11, n, J, jobs)
            for (add, minus, first, count, with_in), o in zip(spec, outs):
                if count == 0:
                    continue
                z = np.zeros((count, Lb), dtype=np.uint64)
                want = oracle.combine(b, pt_l[first:first + count] if with_in else z, masks[add][first:first + count],
                                      masks[minus][first:first + count] if has_minus else z)
                got = o.download(np.uint64, count * Lb).reshape(count, Lb)
                assert np.array_equal(got, want), (b, n, J, has_minus, add, minus, first, count, with_in)


def test_u64_plaintext_zero_extension(E, oracle):
    rng = np.random.Generator(np.random.PCG64(3))
    eng = make(E, 128)
    pt = rng.integers(0, 2 ** 64, 30000, dtype=np.uint64)
    wide = np.stack([pt, np.zeros_like(pt)], axis=1)
    a = eng.encrypt(1, 2, E.SCHEME_DOUBLE, 4, pt)
    assert np.array_equal(a, eng.encrypt(1, 2, E.SCHEME_DOUBLE, 4, wide))
    assert np.array_equal(a, oracle.encrypt(KEY, 1, 2, "double", 4, 128, pt))


def test_counter_beyond_32_bits_and_extreme_prefixes(E, oracle):
    """iter / idx at 2^32 - 1 (FIPS anchors hold the 2^64-1 counter block itself)."""
    eng = make(E, 128)
    it, idx = 2 ** 32 - 1, 2 ** 32 - 2
    pt = np.arange(1000, dtype=np.uint64)
    assert np.array_equal(eng.encrypt(it, idx, E.SCHEME_DOUBLE, 1, pt), oracle.encrypt(KEY, it, idx, "double", 1, 128, pt))
    g = load_golden("aes_anchors.json")
    for c in g["blocks"]:
        assert E.prp_block(KEY, bytes.fromhex(c["block"])).hex() == c["out"]


@pytest.mark.parametrize("b,C,n", [(128, 10, 100003), (128, 1, 17), (64, 3, 100001), (20, 100, 61706),
                                   (128, 100, 61706), (100, 65, 1001), (7, 130, 999)])
def test_aggregate_elem_vs_oracle(E, oracle, b, C, n):
    rng = np.random.Generator(np.random.PCG64(C))
    eng = make(E, b)
    cts = [rand_limbs(rng, n, b) for _ in range(C)]
    assert np.array_equal(eng.aggregate_elem(cts), oracle.aggregate_elem(cts, b))


def test_aggregate_packed_carry_chains(E, oracle):
    """Adversarial carries: all-ones limbs make one +1 ripple across wave, block and pass boundaries."""
    eng = make(E, 64)
    for n_limbs, C in [(1, 2), (2, 3), (511, 2), (512, 2), (513, 5), (1025, 2), (4099, 10), (20000, 66), (3000, 130)]:
        total_bits = n_limbs * 64
        ones = np.full(n_limbs, 2 ** 64 - 1, dtype=np.uint64)
        one = np.zeros(n_limbs, dtype=np.uint64)
        one[0] = 1
        ops = [ones, one] + [np.zeros(n_limbs, dtype=np.uint64) for _ in range(C - 2)]
        got = eng.aggregate_packed(ops, total_bits)
        assert np.array_equal(got, oracle.aggregate_packed(ops, total_bits)), (n_limbs, C)
        assert not got.any()                                            # 2^N - 1 + 1 = 0 mod 2^N
        rng = np.random.Generator(np.random.PCG64(n_limbs))
        ops = [rng.integers(0, 2 ** 64, n_limbs, dtype=np.uint64) for _ in range(C)]
        ops[0][n_limbs // 3: 2 * n_limbs // 3 + 1] = 2 ** 64 - 1        # long propagate run
        ops[1][n_limbs // 3: 2 * n_limbs // 3 + 1] = 0
        for o in ops[2:]:
            o[n_limbs // 3: 2 * n_limbs // 3 + 1] = 0
        for top in (0, 1, 37):
            tb = total_bits - top if n_limbs > 1 or top < 64 else total_bits
            masked = [o.copy() for o in ops]
            assert np.array_equal(eng.aggregate_packed(masked, tb), oracle.aggregate_packed(masked, tb)), (n_limbs, C, top)


def test_packed_slice_helpers(E):
    """flashe_packed_probe_dev / flashe_packed_add_carry_dev (the cross-GPU packed reduce's carry plumbing)
    against Python integers, including ripples through every limb."""
    eng = make(E, 64)
    rng = np.random.Generator(np.random.PCG64(77))
    ones = np.uint64(2 ** 64 - 1)
    for n_limbs in (1, 2, 3, 1023, 1024, 1025, 1026, 5000, 300_001):
        cases = [rng.integers(0, 2 ** 64, n_limbs, dtype=np.uint64), np.full(n_limbs, ones)]
        part = np.full(n_limbs, ones)
        part[n_limbs * 2 // 3] = 5                                     # the ripple stops here
        cases.append(part)
        for x in cases:
            for cin in (0, 1, 7, 2 ** 64 - 1):
                for top in (0, 1, 63):
                    total_bits = 64 * n_limbs - top
                    x = x.copy()
                    if top:
                        x[-1] &= np.uint64((1 << (64 - top)) - 1)
                    d = eng.upload(x)
                    eng.packed_add_carry_dev(n_limbs, total_bits, cin, d)
                    got = int.from_bytes(d.download(np.uint64, n_limbs).tobytes(), "little")
                    want = (int.from_bytes(x.tobytes(), "little") + cin) % (1 << total_bits)
                    assert got == want, (n_limbs, cin, top)
        if n_limbs >= 2:
            info = eng.alloc(24)
            for x in cases:
                for flip in (None, 1, n_limbs - 2, n_limbs - 1, 0):
                    y = x.copy()
                    if flip is not None:
                        y[flip] = 12345
                    d = eng.upload(y)
                    eng.packed_probe_dev(n_limbs, d, info)
                    low, body_ones, top_limb = (int(v) for v in info.download(np.uint64, 3))
                    assert low == int(y[0]) and top_limb == int(y[-1])
                    assert body_ones == int(bool((y[1:n_limbs - 1] == ones).all())), (n_limbs, flip)
    with pytest.raises(Exception):
        eng.packed_probe_dev(1, eng.alloc(16), eng.alloc(24))
    with pytest.raises(Exception):
        eng.packed_add_carry_dev(3, 64, 1, eng.alloc(32))


def test_sharded_round_through_rccl_single_rank():
    """flashe_amd.dist (element-wise, pipelined, fused and packed rounds) with HipOps + a 1-rank RCCL communicator created through
    flashe_rccl_* (no PyTorch in the process); once with the own piece copied locally, once sent through grouped ncclSend / ncclRecv."""
    import subprocess
    import sys
    from conftest import ROOT
    for self_sendrecv in ("0", "1"):
        env = dict(os.environ, FLASHE_RCCL_SELF_SENDRECV=self_sendrecv)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_round_several_ranks_real_kernels(world, tmp_path):
    """The N > 1 round with REAL kernels and `world` ranks: processes share device 0 and run flashe_amd.dist.ShardedRound through
    HipOps exactly as on a multi-GPU node (slices, block-cyclic chunks, sliced decrypts at non-zero `first`, chained job lists,
    device-side packed carry resolution); only the transport is swapped for a file-based double of RcclComm (tests/shm_comm.py),
    because an RCCL group cannot have two ranks on one GPU.  Every schedule, equal / unequal / sparse dealing, vs the oracle.  With
    world = 8 the "uneven" case IS BASELINE config 4's dealing: 10 clients as 2, 2, 1, 1, 1, 1, 1, 1."""
    import subprocess
    import sys
    from conftest import ROOT
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), FLASHE_TEST_SHM_DIR=str(tmp_path), OMP_WAIT_POLICY="passive")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_multi_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {so[-1500:]}{se[-3000:]}"
    assert "DIST_GPU_MULTI_OK" in outs[0][0]


def _bench_shm(tmp_path, extra, env_extra=None, timeout=600):
    """tests/bench_shm.py = bench.py's own main() with 3 ranks on this one GPU over the file-based comm double."""
    import json
    import subprocess
    import sys
    import time
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_shm.py"), "--gpus", "3", "--n", "300007", "--steps", "3", "--warmup", "1",
           "--settle-rounds", "2", "--no-cpu-baseline", "--no-e2e"] + extra
    env = dict(os.environ, OMP_WAIT_POLICY="passive", BENCH_SHM_DIR=str(tmp_path), FLASHE_RDZV_DIR=str(tmp_path))
    env.update(env_extra or {})
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    return r, [json.loads(l) for l in lines], time.time() - t0


def test_bench_multi_rank_sparse_round(tmp_path):
    """`bench.py --config 5 --gpus 3` through the comm double: `value` = every rank its own round (replicas), beside it ONE round shared by
    the ranks by position ranges of the dense vector (SparseShardedRound, parity-gated in-run against the plain sparse sum)."""
    r, lines, _ = _bench_shm(tmp_path, ["--config", "5", "--clients", "6"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 3 and d["value"] > 0 and d["scaling"] == "weak" and "independent replicas" in d["config"]["workload"]
    assert d["value_position_sharded"] > 0 and d["value_position_sharded_no_gather"] > 0 and d["ms_per_step_position_sharded"] > 0


@pytest.mark.parametrize("extra", [["--config", "2", "--clients", "3"], ["--config", "4", "--clients", "5"]])
def test_bench_multi_rank_flow(extra, tmp_path):
    """bench.py's own N > 1 glue -- process spawn, per-rank parity gate with agreement between ranks, the sequential round timed first,
    schedule calibration, barriers and MAX-over-ranks timing, the one JSON line from rank 0 -- run with 3 ranks on this one GPU through
    the file-based comm double (the printed figure is meaningless and labelled so).  Weak scaling (config 2) and the strong-scaling
    deal of config 4."""
    r, lines, _ = _bench_shm(tmp_path, extra)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 3 and d["steps"] == 3 and d["value"] > 0 and "bit-exact" in d["config"]["parity"]
    assert d["scaling"] == ("weak" if extra[1] == "2" else "strong")
    cal = d["config"]["schedule_calibration_ms"]
    assert "sequential" in cal and {k.split(",")[0] for k in cal} == {"fused", "pipelined", "sequential"} and \
        {"fused, 3 chunks", "fused, 4 chunks", "fused, 8 chunks"} <= set(cal)
    assert d["config"]["clients_total"] == (9 if extra[1] == "2" else 5)
    assert d["config"]["ranks_counted_by_allreduce"] == 3 and d["config"]["ranks_parity_ok"] is True
    assert d["config"]["schedule_fallback_reason"] is None and "TEST DOUBLE" in d["config"]["collectives"]
    if d["config"]["schedule_name"] not in ("sequential", "partial-agg"):
        assert d["sequential_ms_per_step"] > 0          # the line that would have been the fallback was measured first
    # both partitions in the one line: `value` = clients sharded over the ranks, beside it the element-sharded round
    # (every rank plays every client on its slice; parity-gated in-run), with and without the all-gather of the decrypted slices
    assert "element_sharded_error" not in d, d.get("element_sharded_error")
    assert d["value_element_sharded"] > 0 and d["value_element_sharded_no_gather"] > 0 and d["ms_per_step_element_sharded"] > 0
    # the line interprets itself: DESIGN section 5's model evaluated with this run's AES rate, and what RCCL saw
    pm, sc = d["predicted_ms"], d["scale_check"]
    cs, es = pm["client_sharded_sequential"], pm["element_sharded"]
    assert abs(cs["round"] - sum(v for k, v in cs.items() if k != "round")) < 1e-9 and cs["all_to_all"] > 0 and cs["encrypt_busiest_rank"] > 0
    assert es["round"] > es["round_no_gather"] > 0 and pm["inputs"]["world"] == 3 and pm["inputs"]["aes_blocks_per_s_measured_this_run"] > 0
    assert pm["inputs"]["clients_per_rank"] == ([3, 3, 3] if extra[1] == "2" else [2, 2, 1])
    assert sc["ranks_counted_by_allreduce_is_n"] is True and sc["ranks_parity_ok"] is True and isinstance(sc["rccl_world_is_n"], bool)
    r_ = sc["measured_over_predicted"]
    assert r_["schedule_that_ran_over_sequential_model"] > 0 and r_["element_sharded_over_model"] > 0 and r_["element_sharded_no_gather_over_model"] > 0
    # the line certifies what produced it: the preflight's view of the machine and the library file's hash
    import hashlib
    from flashe_amd import _lib
    cfg = d["config"]
    assert cfg["devices_visible"] >= 1 and cfg["library"] == os.path.basename(_lib.LIB_PATH) and cfg["abi_version"] == 4
    assert cfg["library_sha256_16"] == hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    assert cfg["peer_access_ok"] is None and cfg["rccl_version"] is None        # (the comm double: several ranks on one device, no RCCL)


def test_bench_multi_rank_allreduce_exchange(tmp_path):
    """bench.py --bits 20 --collective allreduce with 3 ranks through the comm double: the sequential round's exchange is the
    all-reduce form (ncclAllReduce(uint64, sum) + mask for int_bits <= 64), parity on every rank."""
    r, lines, _ = _bench_shm(tmp_path, ["--config", "2", "--clients", "3", "--bits", "20", "--collective", "allreduce", "--schedule", "sequential"])
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    d = lines[0]
    assert d["n_gpus"] == 3 and d["value"] > 0 and d["config"]["ranks_parity_ok"] is True and "ncclAllReduce" in d["config"]["exchange"]
    r, lines, _ = _bench_shm(tmp_path, ["--config", "2", "--clients", "3", "--collective", "allreduce", "--schedule", "sequential"])
    assert r.returncode != 0 and not [l for l in lines if l.get("value")]              # 128-bit modulus: refused


@pytest.mark.parametrize("inject", ["raise:1", "hang:2", "raise:0"])
def test_bench_multi_rank_falls_back_to_the_sequential_line(inject, tmp_path):
    """The first real multi-GPU run must not come back empty: with an overlapped schedule that raises on one rank, or blocks one
    rank forever (the others then sit in their collective), every rank leaves at the deadline at the latest and rank 0 prints the
    sequential line it had already measured, exit code 0, `config.schedule_fallback_reason` saying why."""
    r, lines, took = _bench_shm(tmp_path, ["--config", "2", "--clients", "3", "--calibration-deadline", "25"], {"BENCH_SHM_INJECT": inject}, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = lines[0]
    # (with an exchange the sequential round sends the partial aggregate the encrypt launch wrote: schedule_name "partial-agg")
    assert d["value"] > 0 and d["n_gpus"] == 3 and d["config"]["schedule_name"] in ("sequential", "partial-agg")
    why = d["config"]["schedule_fallback_reason"]
    assert why and (("injected failure" in why) if inject.startswith("raise") else ("deadline" in why)), why
    assert took < 200, took


@pytest.mark.parametrize("inject", ["raise_elements:1", "hang_elements:2"])
def test_bench_multi_rank_keeps_its_line_when_the_element_sharded_phase_fails(inject, tmp_path):
    """The element-sharded partition is timed AFTER the client-sharded line exists: a rank that raises or hangs there (the others
    then sit in a collective) must not cost the run its result -- rank 0 prints the main line, exit code 0, within the deadline."""
    r, lines, took = _bench_shm(tmp_path, ["--config", "4", "--clients", "5", "--schedule", "sequential", "--calibration-deadline", "25"],
                                {"BENCH_SHM_INJECT": inject}, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = lines[0]
    assert d["value"] > 0 and d["n_gpus"] == 3 and d["config"]["ranks_parity_ok"] is True
    assert "value_element_sharded" not in d and d["config"]["schedule_fallback_reason"] is None
    assert "element_sharded_error" in d and (("injected failure" in d["element_sharded_error"]) if inject.startswith("raise") else ("deadline" in d["element_sharded_error"]))
    assert took < 200, took


def test_bench_partial_agg_schedule_multi_rank(tmp_path):
    """--schedule partial-agg with 3 ranks: the encrypt launch writes each rank's partial aggregate, which is what the exchange sends."""
    r, lines, _ = _bench_shm(tmp_path, ["--config", "2", "--clients", "3", "--schedule", "partial-agg"])
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    assert lines[0]["config"]["schedule_name"] == "partial-agg" and lines[0]["value"] > 0 and lines[0]["config"]["ranks_parity_ok"] is True


def test_cu_limit_changes_the_launch_shape_not_the_result(E, oracle):
    """flashe_ctx_set_cu_limit (CUs left free for RCCL's transfer kernels): the persistent launches fill fewer CUs, every result
    stays bit-identical -- chained wide launch, b <= 64 chain, reduce fused with the decrypt, a one-CU extreme."""
    n, C = 300_007, 5
    for b in (128, 64):
        eng = make(E, b)
        assert eng.cu_count >= 64
        Lb = L(b)
        rng = np.random.Generator(np.random.PCG64(b))
        pts = [rng.integers(0, 2 ** 63, n, dtype=np.uint64) for _ in range(C)]
        dp = [eng.upload(p) for p in pts]
        want = [oracle.encrypt(KEY, 3, c, "double", 16, b, pts[c]) for c in range(C)]
        for limit in (eng.cu_count - 32, 7, 1, 0, 10 ** 6):
            eng.set_cu_limit(limit)
            dc = [eng.alloc_vec(n) for _ in range(C)]
            eng.encrypt_batch_dev(3, list(range(C)), E.SCHEME_DOUBLE, n, 16, dp, 1, dc)
            for c in range(C):
                assert np.array_equal(dc[c].download(np.uint64, n * Lb).reshape(n, Lb), want[c]), (b, limit, c)
            out = eng.alloc_vec(n)
            eng.decrypt_dev(3, [C], [0], n, 16, dc[0], out)
            assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), oracle.decrypt(KEY, 3, [C], [0], 16, b, want[0])), (b, limit)
        with pytest.raises(E.FlasheError):
            eng.set_cu_limit(-1)


def test_contexts_are_independent_across_threads(E, oracle):
    """One ctx is not thread-safe, different ctxs are (include/flashe.h): four threads, each with its own engine (own stream, key
    and bit width), run host-pointer and device calls concurrently -- ctypes releases the GIL during every call -- and every
    result equals the oracle's; the shared pieces underneath (host result pool, library-level statics) must hold."""
    import threading
    errors = []

    def worker(t):
        try:
            key = bytes((17 * t + i) & 255 for i in range(32))
            b = (128, 64, 100, 23)[t]
            eng = E.Engine(key, b, device=0)
            rng = np.random.Generator(np.random.PCG64(900 + t))
            for rep in range(12):
                n = int(rng.integers(1, 200_000))
                pt = rng.integers(0, 2 ** min(b, 63), n, dtype=np.uint64)
                ct = eng.encrypt(rep, t, E.SCHEME_DOUBLE, 16, pt)
                assert np.array_equal(ct, oracle.encrypt(key, rep, t, "double", 16, b, pt)), (t, rep, "encrypt")
                dec = eng.decrypt(rep, [t + 1], [t], 16, ct)
                assert np.array_equal(dec[:, 0], pt), (t, rep, "decrypt")
                d = eng.upload(ct)
                out = eng.alloc_vec(n)
                eng.aggregate_elem_dev([d, d, d], n, out)
                assert np.array_equal(out.download(np.uint64, ct.size).reshape(ct.shape), oracle.aggregate_elem([ct, ct, ct], b)), (t, rep, "reduce")
        except Exception as exc:          # pragma: no cover - reported below
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors


def test_c_example_round_runs(tmp_path):
    """The drop-in boundary from a non-Python host: examples/c_round.c (plain C11 against include/flashe.h) encrypts four clients'
    vectors, reduces, telescopes the prefix list and decrypts -- the result must be the plain sum."""
    import subprocess
    from test_abi_and_host import _build_c_example
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_ROUND OK" in r.stdout and "C_ROUND_U32 OK" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
    # third round: BASELINE config 3's shape (LeNet-sized vector, 100 clients, mask precompute held and consumed inside the ctx; a
    # dropout whose two uncovered prefixes are computed online) -- VERDICT r3 #9
    assert "C_ROUND_PRECOMPUTE OK" in r.stdout and "clients=100" in r.stdout and "extra_prefixes=1+1" in r.stdout, r.stdout[-800:]
    # fourth round: config 5's path, device resident -- six sparse clients encrypted and summed in one call, the sparse decrypt, shared span bounds
    assert "C_ROUND_SPARSE OK" in r.stdout and "ciphertext_differences=0" in r.stdout, r.stdout[-800:]


@pytest.mark.parametrize("b,n,J", [(128, 61_706, 16), (23, 61_706, 16), (64, 5_001, 3), (128, 0, 1), (20, 7, 16)])
def test_ctx_resident_precompute_vs_oracle(E, oracle, b, n, J):
    """flashe_prepare_encrypt / flashe_prepare_decrypt + flashe_*_prepared_dev (jzf_flashe.py:599-666): the cached masks equal the
    oracle's streams, the prepared encrypt equals the online one, the cache is consumed by ONE call and refuses the next, a length
    mismatch leaves it in place (NumPy's broadcast error does not delete the reference's cache either), extras of a dropout are merged
    in online, and the host twins give the same bytes."""
    from flashe_amd._lib import FlasheError
    eng = make(E, b)
    Lb = L(b)
    C, it, idx = 5, 9, 2
    rng = np.random.Generator(np.random.PCG64(b + n))
    pt = rng.integers(0, 2 ** min(b, 60), n, dtype=np.uint64)
    dp = eng.upload(pt) if n else eng.alloc(16)
    out = eng.alloc_vec(max(n, 1))
    assert eng.prepared_query(eng.PREPARED_ENCRYPT)[0] is False
    eng.prepare_encrypt(it, idx, 1, n, J)
    held, nn, a_ptr, m_ptr = eng.prepared_query(eng.PREPARED_ENCRYPT)
    assert held and nn == n and (n == 0 or (a_ptr and m_ptr))
    if n:
        assert np.array_equal(eng.prepared_download(eng.PREPARED_ENCRYPT, "add"), oracle.mask_sum(KEY, it, [idx], n, J, b))
        assert np.array_equal(eng.prepared_download(eng.PREPARED_ENCRYPT, "minus"), oracle.mask_sum(KEY, it, [idx + 1], n, J, b))
        with pytest.raises(FlasheError):
            eng.encrypt_prepared_dev(n + 1, dp, 1, out)              # wrong length: refused, cache kept
        assert eng.prepared_query(eng.PREPARED_ENCRYPT)[0] is True
    eng.encrypt_prepared_dev(n, dp, 1, out)
    want_ct = oracle.encrypt(KEY, it, idx, "double", J, b, pt)
    assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), want_ct)
    assert eng.prepared_query(eng.PREPARED_ENCRYPT)[0] is False      # consumed
    with pytest.raises(FlasheError):
        eng.encrypt_prepared_dev(n, dp, 1, out)
    # decrypt: all C uploaded (everything precomputed), then client 1 dropped (two extra prefixes online)
    cts = [oracle.encrypt(KEY, it, c, "double", J, b, rng.integers(0, 2 ** min(b, 60), n, dtype=np.uint64)) for c in range(C)]
    for up in (list(range(C)), [0, 2, 3, 4]):
        agg = oracle.aggregate_elem([cts[c] for c in up], b) if n else np.zeros((0, Lb), dtype=np.uint64)
        add, minus = E.telescope(sorted(up))
        want = oracle.decrypt(KEY, it, add, minus, J, b, agg)
        xa, xm = [i for i in add if i != C], [i for i in minus if i != 0]
        da = eng.upload(agg) if n else eng.alloc(16)
        eng.prepare_decrypt(it, C, n, J)
        eng.decrypt_prepared_dev(it, xa, xm, n, J, da, out)
        assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), want), (b, n, up)
        assert eng.prepared_query(eng.PREPARED_DECRYPT)[0] is False
    # single mask; discard; host twins
    eng.prepare_encrypt(it, idx, 0, n, J)
    assert eng.prepared_query(eng.PREPARED_ENCRYPT)[3] is None       # no minus vector
    eng.encrypt_prepared_dev(n, dp, 1, out)
    assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), oracle.encrypt(KEY, it, idx, "single", J, b, pt))
    eng.prepare_encrypt(it, idx, 1, n, J)
    eng.prepared_discard(eng.PREPARED_ENCRYPT | eng.PREPARED_DECRYPT)
    assert eng.prepared_query(eng.PREPARED_ENCRYPT)[0] is False
    if n:
        import ctypes
        lib = eng._lib
        host_ct = np.zeros((n, Lb), dtype=np.uint64)
        eng.prepare_encrypt(it, idx, 1, n, J)
        eng._check(lib.flashe_encrypt_prepared(eng._h, n, pt.ctypes.data, 1, host_ct.ctypes.data))
        assert np.array_equal(host_ct, want_ct)
        agg = oracle.aggregate_elem(cts, b)
        host_out = np.zeros((n, Lb), dtype=np.uint64)
        eng.prepare_decrypt(it, C, n, J)
        eng._check(lib.flashe_decrypt_prepared(eng._h, it, None, 0, None, 0, n, J, agg.ctypes.data, host_out.ctypes.data))
        assert np.array_equal(host_out, oracle.decrypt(KEY, it, [C], [0], J, b, agg))
        del ctypes


def test_staging_pool_under_a_tight_budget():
    """The host-pointer calls keep their device staging blocks (abi.hip `Tmp`) within FLASHE_STAGING_POOL_MB: with 3 MB allowed,
    a mix of sizes forces reuse, eviction of parked blocks, slots emptied in place and plain allocations for what does not
    fit -- every result still equals the oracle's, and the recycled host result arrays (engine._HostPool, 2 MB allowed) never
    alias while alive.  Own process: both budgets are read once."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from flashe_amd.engine import Engine, SCHEME_DOUBLE
from oracle import flashe_oracle as orc
KEY = bytes(range(32))
eng = Engine(KEY, 128)
rng = np.random.Generator(np.random.PCG64(5))
sizes = [1000, 90_000, 3, 250_000, 40_000, 90_000, 131_072, 1, 250_000, 70_000, 1000, 200_000]
alive = []
for rep in range(3):
    for i, n in enumerate(sizes):
        pt = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
        ct = eng.encrypt(rep, i, SCHEME_DOUBLE, 16, pt)
        assert np.array_equal(ct, orc.encrypt(KEY, rep, i, "double", 16, 128, pt)), ("encrypt", rep, n)
        dec = eng.decrypt(rep, [i + 1], [i], 16, ct)
        assert np.array_equal(dec[:, 0], pt) and not dec[:, 1].any(), ("decrypt", rep, n)
        agg = eng.aggregate_elem([ct, ct, dec])
        assert np.array_equal(agg, orc.aggregate_elem([ct, ct, dec], 128)), ("aggregate", rep, n)
        alive.append((ct, ct.copy()))
        if len(alive) > 4:
            alive.pop(0)
        for a, c in alive:
            assert np.array_equal(a, c), "a live result array was overwritten"
print("POOL_OK")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, FLASHE_STAGING_POOL_MB="3", FLASHE_HOST_POOL_MB="2", OMP_WAIT_POLICY="passive"))
    assert r.returncode == 0 and "POOL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("b,n", [(128, 50001), (120, 40003), (65, 999), (64, 70001), (33, 12345), (23, 61706),
                                 (20, 100000), (8, 4097), (7, 30000), (1, 10000)])
def test_pack_unpack_vs_oracle(E, oracle, b, n):
    rng = np.random.Generator(np.random.PCG64(b))
    eng = make(E, b)
    x = rand_limbs(rng, n, b)
    p = eng.pack(x)
    assert np.array_equal(p, oracle.pack(x, b))
    assert np.array_equal(eng.unpack(p, n), x)
    ys = [rand_limbs(rng, n, b) for _ in range(4)]
    got = eng.unpack(eng.aggregate_packed([eng.pack(y) for y in ys], n * b), n)
    want = oracle.unpack(oracle.aggregate_packed([oracle.pack(y, b) for y in ys], n * b), n, b)
    assert np.array_equal(got, want)


def test_sparse_config5_shape_small(E, oracle):
    """BASELINE config 5 scaled down: total = 255 570, 1 % per client, C = 8, b = 128, single mask."""
    total, C, b, J, it = 255570, 8, 128, 16, 2
    rng = np.random.Generator(np.random.PCG64(2000))
    eng = make(E, b)
    locs = [np.sort(rng.choice(total, size=total // 100, replace=False)).astype(np.uint32) for _ in range(C)]
    dense, zeros = [], []
    for c in range(C):
        pt = rng.integers(0, 2 ** 60, len(locs[c]), dtype=np.uint64)
        ct = eng.encrypt(it, c, E.SCHEME_SINGLE, J, pt)
        z = np.array([int(rng.integers(0, 2 ** 15)), 0], dtype=np.uint64)
        zeros.append(z)
        d = eng.expand_to_dense(total, locs[c], ct, z)
        assert np.array_equal(d, oracle.expand_to_dense(total, locs[c], ct, z, b))
        dense.append(d)
    agg = eng.aggregate_elem(dense)
    mm = eng.sparse_minus_mask(it, locs, total, J)
    assert np.array_equal(mm, oracle.sparse_minus_mask(KEY, it, locs, total, J, b))
    dec = eng.combine(agg, None, mm)
    assert np.array_equal(dec, oracle.combine(b, oracle.aggregate_elem(dense, b), None, mm))


@pytest.mark.parametrize("b,total,C", [(128, 50_000, 7), (100, 4099, 3), (64, 30_001, 5), (20, 9999, 9), (128, 64, 4), (128, 300_000, 70), (128, 2048, 2),
                                       (128, 2049, 2), (128, 4096, 2), (128, 4097, 3), (64, 8193, 4)])
def test_sparse_aggregate_fused_vs_dense_path(E, oracle, b, total, C):
    """flashe_sparse_aggregate_dev == expand_to_dense per client + element-wise reduce (overlapping location sets,
    full-range zero values, an empty client, a client that covers every position)."""
    rng = np.random.Generator(np.random.PCG64(total + C))
    eng = make(E, b)
    Lb = L(b)
    ks = [int(v) for v in rng.integers(1, max(2, total // 3), C)]
    ks[0], ks[-1] = 0, total
    locs = [np.sort(rng.choice(total, size=k, replace=False)).astype(np.uint32) for k in ks]
    vals = [rand_limbs(rng, max(k, 1), b)[:k] for k in ks]
    zeros = [rand_limbs(rng, 1, b)[0] for _ in range(C)]
    dense = [oracle.expand_to_dense(total, locs[c], vals[c], zeros[c], b) for c in range(C)]
    want = oracle.aggregate_elem(dense, b)
    dl = [eng.upload(l if len(l) else np.zeros(1, dtype=np.uint32)) for l in locs]
    dv = [eng.upload(v if len(v) else np.zeros((1, Lb), dtype=np.uint64)) for v in vals]
    out = eng.alloc_vec(total)
    for sorted_lists in (False, True):
        out.upload(np.full(total * Lb, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64))      # stale contents must not matter
        eng.sparse_aggregate_dev(total, dl, ks, dv, zeros, out, sorted_lists=sorted_lists)
        assert np.array_equal(out.download(np.uint64, total * Lb).reshape(total, Lb), want), sorted_lists
    # unsorted (shuffled) lists through the general form
    perm = [rng.permutation(k) for k in ks]
    dl2 = [eng.upload(l[p] if len(l) else np.zeros(1, dtype=np.uint32)) for l, p in zip(locs, perm)]
    dv2 = [eng.upload(v[p] if len(v) else np.zeros((1, Lb), dtype=np.uint64)) for v, p in zip(vals, perm)]
    eng.sparse_aggregate_dev(total, dl2, ks, dv2, zeros, out, sorted_lists=False)
    assert np.array_equal(out.download(np.uint64, total * Lb).reshape(total, Lb), want)
    # the clients' dense minus-mask: sorted one-pass form == general form == oracle
    mm = oracle.sparse_minus_mask(KEY, 4, locs, total, 16, b)
    for sorted_lists in (False, True):
        eng.sparse_minus_mask_dev(4, dl, ks, total, 16, out, sorted_lists=sorted_lists)
        assert np.array_equal(out.download(np.uint64, total * Lb).reshape(total, Lb), mm), sorted_lists
    # the single-mask sparse decrypt in the pass that builds the mask: out = aggregate - minus-mask (jzf_flashe.py:531-532)
    agg_in = rand_limbs(rng, total, b)
    d_agg = eng.upload(agg_in)
    want_dec = oracle.combine(b, agg_in, None, mm)
    for sorted_lists in (False, True):
        out.upload(np.full(total * Lb, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64))
        eng.sparse_decrypt_dev(4, dl, ks, total, 16, d_agg, out, sorted_lists=sorted_lists)
        assert np.array_equal(out.download(np.uint64, total * Lb).reshape(total, Lb), want_dec), ("fused decrypt", sorted_lists)
        assert np.array_equal(d_agg.download(np.uint64, total * Lb).reshape(total, Lb), agg_in)       # the aggregate is only read
    with pytest.raises(E.FlasheError):
        eng.sparse_decrypt_dev(4, dl, ks, total, 16, out, out, sorted_lists=True)                    # in place is refused
    if b not in (64, 128):
        with pytest.raises(Exception):                     # a zero value wider than int_bits
            eng.sparse_aggregate_dev(total, dl[:1], ks[:1], dv[:1], [[2 ** 64 - 1, 2 ** 64 - 1]], out)


# ------------------------------------------------------------------ quantise / batch codec (8f-1)
def test_codec_golden(E, oracle):
    g = load_golden("codec.json")
    e64 = make(E, 64)
    for c in g["quantize"]:
        x = np.frombuffer(bytes.fromhex(c["x"]), dtype=c["dtype"])
        u = np.frombuffer(bytes.fromhex(c["u"]), dtype=np.float64)
        q = e64.quantize(x, float.fromhex(c["alpha"]), c["element_bits"], u)
        assert [int(v) for v in q] == unhex(c["q"]), (c["dtype"], c["element_bits"])
    for c in g["batch"]:
        eng = make(E, c["int_bits"])
        fb = c["element_bits"] + c["factor"]
        b = eng.batch(np.array(unhex(c["vals"]), dtype=np.uint64), fb)
        assert oracle.limbs_to_ints(b) == unhex(c["batched"])
        assert [int(v) for v in eng.unbatch(b, fb)] == unhex(c["unbatched"])
    e128 = make(E, 128)
    for c in g["unquantize"]:
        want = np.frombuffer(bytes.fromhex(c["out"]), dtype=np.float64)
        got = e128.unquantize(oracle.ints_to_limbs(unhex(c["vals"]), 128), float.fromhex(c["alpha"]), c["element_bits"], c["num_clients"])
        assert got.tobytes() == want.tobytes()


def test_codec_vs_oracle_large(E, oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    e64, e128 = make(E, 64), make(E, 128)
    for dtype, bits in ((np.float32, 16), (np.float32, 32), (np.float64, 24)):
        x = (rng.standard_normal(300001) * 4).astype(dtype)
        u = rng.random(300001)
        assert np.array_equal(e64.quantize(x, 8.17121, bits, u), oracle.quantize(x, 8.17121, bits, u))
    v = rng.integers(0, 2 ** 64, size=(200001, 2), dtype=np.uint64)
    v[::3, 1] = 0
    v[1::7, 1] &= np.uint64(0xFF)
    assert e128.unquantize(v, 6.5, 32, 10).tobytes() == oracle.unquantize(v, 6.5, 32, 10).tobytes()
    for int_bits, fb in ((128, 20), (120, 20), (64, 17), (20, 20), (100, 33)):
        eng = make(E, int_bits)
        vals = rng.integers(0, 2 ** fb, 100003, dtype=np.uint64)
        b = eng.batch(vals, fb)
        assert np.array_equal(b, oracle.batch(vals, int_bits, fb))
        assert np.array_equal(eng.unbatch(b, fb), oracle.unbatch(b, int_bits, fb))


def test_config1_end_to_end_with_codec(E):
    """BASELINE config 1 with the codec on the GPU too: fp32 -> quantise -> encrypt -> aggregate ->
    decrypt -> unquantise, every stage equal to the reference's fixture (same MT19937 draws)."""
    from flashe_amd import quantize as qz
    z = np.load(os.path.join(GOLDEN, "config1.npz"))
    n, b, J, alpha = 10000, 64, 8, float(z["alpha"])
    eng = make(E, b)
    cts = []
    for c in range(2):
        np.random.seed(7 + c)
        q = qz._static_quantize_padding_asymmetric(z[f"x{c}"], alpha, 32, as_object=False)
        assert np.array_equal(q, z[f"q{c}"])
        ct = eng.encrypt(0, c, E.SCHEME_SINGLE, J, q)
        assert np.array_equal(ct[:, 0], z[f"ct{c}"])
        cts.append(ct)
    dec = eng.decrypt(0, [], [0, 1], J, eng.aggregate_elem(cts))
    unq = qz._static_unquantize_padding_asymmetric(dec[:, 0], alpha, 32, 2)
    assert unq.tobytes() == z["unq_elem"].tobytes()
    unq_obj = qz._static_unquantize_padding_asymmetric(dec[:, 0].astype(object), alpha, 32, 2)
    assert unq_obj.tobytes() == z["unq_elem"].tobytes()


def test_wire_format_mirror(E):
    """f-2: compress / decompress integers equal the reference's (_to_bytes_old fixtures, incl. the chunk-merge)."""
    from flashe_amd import weights as wz
    g = load_golden("pack.json")
    for c in g["cases"] + g["merges"]:
        vals = unhex(c["vals"])
        big, n = wz.to_big_int(np.array(vals, dtype=object), c["b"])
        assert n == c["n"] and big == int(c["packed_int"], 16)
        assert [int(v) for v in wz.from_big_int(big, n, c["b"])] == vals
        # device-resident ends of the wire format: a handle is packed where it lies, a received integer is unpacked into HBM
        if n:
            from oracle.flashe_oracle import limbs_to_ints
            dv = wz.from_big_int(big, n, c["b"], as_device=True)
            assert isinstance(dv, E.DeviceVector) and limbs_to_ints(dv.to_host()) == vals
            assert wz.to_big_int(dv, c["b"]) == (big, n)
    layer = np.array([[3, 1, 4], [1, 5, 9]], dtype=object)
    tw = wz.TransferableWeights({"w": layer, "b": np.array([7], dtype=object)}, bits=20)
    assert tw.unboxed["w"] == sum(int(v) << (20 * (5 - j)) for j, v in enumerate(layer.flatten()))
    back = tw.decompress()
    assert back["w"].shape == (2, 3) and [int(v) for v in back["w"].flatten()] == [3, 1, 4, 1, 5, 9] and int(back["b"][0]) == 7
    dev_layers = wz.TransferableWeights({"w": tw.unboxed["w"]}, bits=20, need_compress=False, shape={"w": (2, 3)}).decompress(as_device=True)
    assert isinstance(dev_layers["w"], E.DeviceVector)
    again = wz.TransferableWeights({"w": dev_layers["w"]}, bits=20)                     # a handle compresses to the same integer
    assert again.unboxed["w"] == tw.unboxed["w"]
    # sparse location coding: _to_bytes(locations, total.bit_length())
    total = 25_557_032
    locs = np.sort(np.random.Generator(np.random.PCG64(1)).choice(total, size=5000, replace=False))
    big, n = wz.to_big_int(locs.astype(object), total.bit_length())
    assert [int(v) for v in wz.from_big_int(big, n, total.bit_length())] == [int(v) for v in locs]


def test_sparsify_golden_and_oracle(E, oracle):
    """f-3: top-k + residual, the reference's fixtures (two rounds) and a config-5-sized layer vs the oracle."""
    eng = make(E, 128)
    for c in load_golden("sparsify.json")["cases"]:
        dt = np.dtype(c["dtype"])
        remain = np.zeros(c["n"], dtype=dt)
        for rd in c["rounds"]:
            layer = np.frombuffer(bytes.fromhex(rd["layer"]), dtype=dt)
            loc, vals, remain = eng.sparsify(layer, rd["k"], remain)
            assert [int(v) for v in loc] == rd["location"]
            assert vals.tobytes().hex() == rd["masked"] and remain.tobytes().hex() == rd["remain"]
    rng = np.random.Generator(np.random.PCG64(8))
    for dt, n, k in [(np.float32, 2_555_703, 25_557), (np.float64, 300_001, 1), (np.float32, 5000, 5000), (np.float32, 70_000, 69_999)]:
        layer = rng.standard_normal(n).astype(dt)
        layer[::97] = layer[5]                       # plenty of exact ties, some at the threshold for small k
        res = rng.standard_normal(n).astype(dt)
        loc, vals, new = eng.sparsify(layer, k, res)
        wl, wv, wr = oracle.sparsify(layer, k, res)
        assert np.array_equal(loc, wl) and vals.tobytes() == wv.tobytes() and new.tobytes() == wr.tobytes(), (dt, n, k)
    loc, vals, none = eng.sparsify(np.array([1, -2, 2, 0.5, 2, -1], dtype=np.float32), 2)
    assert [int(v) for v in loc] == [2, 4] and none is None
    # the dict-level mirror: two layers, locations bit-packed with total.bit_length() bits
    from flashe_amd import weights as wz
    sp = wz.Sparsifier(0.1)
    w = {"a": rng.standard_normal((30, 10)).astype(np.float32), "b": rng.standard_normal(50).astype(np.float32)}
    ref = {k: v.copy() for k, v in w.items()}
    enc, le, bits, total = sp.sparsify(w)
    assert (le, bits, total) == (30 + 5, (350).bit_length(), 350) and w["a"].shape == (30,) and w["b"].shape == (5,)
    locs = [int(v) for v in wz.from_big_int(enc, le, bits)]
    want_a = sorted(np.abs(ref["a"].flatten()).argsort(kind="stable")[-30:].tolist())
    want_b = sorted((np.abs(ref["b"]).argsort(kind="stable")[-5:] + 300).tolist())
    assert locs == want_a + want_b
    assert np.array_equal(w["a"], ref["a"].flatten()[want_a]) and sp.remain_weights["a"][want_a].sum() == 0


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_sparsify_every_layer_in_one_set_of_launches(E, oracle, dt):
    """flashe_sparsify_batch[_dev] (Client.sparsify's layer loop, jzf_aggregator.py:585-613, as one set of launches): layer by layer
    the same locations, values and residuals as the one-layer call and as the oracle -- layers of 1 .. 2.4 M elements, block-sized
    and off-by-one sizes, k = 1 .. n, ties at the threshold, with and without residuals, a second round on the updated residuals;
    and the dict-level Sparsifier on a model of 40 layers."""
    eng = make(E, 128)
    rng = np.random.Generator(np.random.PCG64(17))
    sizes = [1, 2, 5, 150, 1023, 1024, 1025, 2048, 2400, 48_000, 61_706, 100_000, 2_359_296, 7, 4096, 3, 999_999]
    layers, ks, res = [], [], []
    for i, n in enumerate(sizes):
        l = rng.standard_normal(n).astype(dt)
        if n > 10:
            l[::7] = l[3]                                 # many exact ties, often at the threshold
        layers.append(l)
        ks.append(int(max(1, [n // 100, n, 1, n - 1, n // 2][i % 5])))
        res.append(rng.standard_normal(n).astype(dt))
    for rnd in range(2):
        got = eng.sparsify_batch(layers, ks, res)
        for l, k, r, (loc, vals, new) in zip(layers, ks, res, got):
            wl, wv, wr = oracle.sparsify(l, k, r)
            assert np.array_equal(loc, wl) and vals.tobytes() == wv.tobytes() and new.tobytes() == wr.tobytes(), (dt, rnd, l.size, k)
        res = [g[2].copy() for g in got]                                    # next round: the residuals this one left
        layers = [rng.standard_normal(l.size).astype(dt) for l in layers]
    got = eng.sparsify_batch(layers[:5], ks[:5], None)                     # no residuals
    for l, k, (loc, vals, new) in zip(layers[:5], ks[:5], got):
        wl, wv, _ = oracle.sparsify(l, k, np.zeros_like(l))
        assert np.array_equal(loc, wl) and vals.tobytes() == wv.tobytes() and new is None
    # device form on one flat buffer == host form
    flat = np.concatenate(layers)
    dx, dr = eng.upload(flat), eng.upload(np.concatenate(res))
    dl, dv = eng.alloc(4 * sum(ks) + 16), eng.alloc(flat.itemsize * sum(ks) + 16)
    eng.sparsify_batch_dev([l.size for l in layers], ks, dx, dt == np.float64, dr, dl, dv)
    want = eng.sparsify_batch(layers, ks, res)
    assert np.array_equal(dl.download(np.uint32, sum(ks)), np.concatenate([w[0] for w in want]))
    assert dv.download(dt, sum(ks)).tobytes() == np.concatenate([w[1] for w in want]).tobytes()
    assert dr.download(dt, flat.size).tobytes() == np.concatenate([w[2] for w in want]).tobytes()
    with pytest.raises(E.FlasheError):
        eng.sparsify_batch_dev([10, 20], [5, 21], dx, dt == np.float64, dr, dl, dv)      # k > n
    # the dict-level mirror takes the batch path and equals the layer-by-layer walk
    from flashe_amd import weights as wz
    model = {f"l{i:02d}": rng.standard_normal(int(s)).astype(dt) for i, s in enumerate(rng.integers(1, 5000, 40))}
    a, b = wz.Sparsifier(0.05), wz.Sparsifier(0.05)
    real = E.Engine.sparsify_model
    for rnd in range(3):                                      # the residuals of one round feed the next: on the device for a, on the host for b
        model = {k: rng.standard_normal(v.size).astype(dt) for k, v in model.items()}
        wa, wb = {k: v.copy() for k, v in model.items()}, {k: v.copy() for k, v in model.items()}
        ea = a.sparsify(wa)
        try:
            del E.Engine.sparsify_model                       # force the layer-by-layer walk
            eb = b.sparsify(wb)
        finally:
            E.Engine.sparsify_model = real
        assert ea == eb and all(wa[k].tobytes() == wb[k].tobytes() for k in model), rnd
        if rnd != 1:                                          # (round 1: a's residuals are never downloaded)
            assert all(a.remain_weights[k].tobytes() == b.remain_weights[k].tobytes() for k in model), rnd
    # a model whose layers change shape between rounds keeps the residuals of the layers it still has
    c, d = wz.Sparsifier(0.05), wz.Sparsifier(0.05)
    m1 = {"x": rng.standard_normal(3000).astype(dt), "y": rng.standard_normal(500).astype(dt)}
    m2 = {"x": rng.standard_normal(3000).astype(dt), "z": rng.standard_normal(70).astype(dt)}
    for sp_, forced in ((c, False), (d, True)):
        for m in (m1, m2):
            w = {k: v.copy() for k, v in m.items()}
            if forced:
                try:
                    del E.Engine.sparsify_model
                    sp_.sparsify(w)
                finally:
                    E.Engine.sparsify_model = real
            else:
                sp_.sparsify(w)
    assert all(c.remain_weights[k].tobytes() == d.remain_weights[k].tobytes() for k in ("x", "z"))
    a.remain_weights = {k: v.copy() for k, v in b.remain_weights.items()}        # assigned from outside: uploaded again
    wa, wb = {k: v.copy() for k, v in model.items()}, {k: v.copy() for k, v in model.items()}
    assert a.sparsify(wa) == b.sparsify(wb) and all(a.remain_weights[k].tobytes() == b.remain_weights[k].tobytes() for k in model)


# ------------------------------------------------------------------ BASELINE full sizes
def _sum_u64(pts):
    lo = np.zeros_like(pts[0])
    hi = np.zeros_like(pts[0])
    for p in pts:
        new = lo + p
        hi += (new < lo).astype(np.uint64)
        lo = new
    return lo, hi


def test_config2_full_size_round(E, oracle):
    """n = 1e7, b = 128, C = 10, double mask, data resident in HBM: the decrypted aggregate must equal
    the plaintext sum (encrypt -> aggregate -> decrypt round trip), a dropout round must too, and one
    client's full ciphertext plus the full aggregate are compared with the oracle."""
    n, C, b, it = 10_000_000, 10, 128, 0
    eng = make(E, b)
    pts = [np.random.Generator(np.random.PCG64(1000 + c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in range(C)]
    # what bench.py runs: ONE chained launch for the ten clients (11 shared PRF streams instead of 20)
    eng.encrypt_batch_dev(it, list(range(C)), E.SCHEME_DOUBLE, n, 16, dpt, 1, dct)
    dagg, dout = eng.alloc_vec(n), eng.alloc_vec(n)
    eng.aggregate_elem_dev(dct, n, dagg)
    eng.decrypt_dev(it, [C], [0], n, 16, dagg, dout)
    out = dout.download(np.uint64, 2 * n).reshape(n, 2)
    lo, hi = _sum_u64(pts)
    assert np.array_equal(out[:, 0], lo) and np.array_equal(out[:, 1], hi)
    # EVERY client's full 1e7-element ciphertext against the oracle, which encrypts each client on its own
    for c in range(C):
        ct = dct[c].download(np.uint64, 2 * n).reshape(n, 2)
        assert np.array_equal(ct, oracle.encrypt(KEY, it, c, "double", 16, b, pts[c])), c
    # and the same ciphertexts from ten separate launches
    d_one = eng.alloc_vec(n)
    eng.encrypt_dev(it, 3, E.SCHEME_DOUBLE, n, 16, dpt[3], 1, d_one)
    assert np.array_equal(d_one.download(np.uint64, 2 * n), dct[3].download(np.uint64, 2 * n))
    # dropout: clients 4 and 7 missing -> telescoped prefixes
    up = [0, 1, 2, 3, 5, 6, 8, 9]
    eng.aggregate_elem_dev([dct[c] for c in up], n, dagg)
    add_idx, minus_idx = E.telescope(list(up))
    assert (add_idx, minus_idx) == ([4, 7, 10], [0, 5, 8])
    eng.decrypt_dev(it, add_idx, minus_idx, n, 16, dagg, dout)
    out = dout.download(np.uint64, 2 * n).reshape(n, 2)
    lo, hi = _sum_u64([pts[c] for c in up])
    assert np.array_equal(out[:, 0], lo) and np.array_equal(out[:, 1], hi)
    # packed reduce at full size equals element-wise + carry-ins: check via oracle on the packed ints
    dpk = [eng.alloc_vec(n) for _ in range(3)]
    for k in range(3):
        eng.pack_dev(n, dct[k], dpk[k])
    dsum = eng.alloc_vec(n)
    eng.aggregate_packed_dev(dpk, 2 * n, 128 * n, dsum)
    got = dsum.download(np.uint64, 2 * n)
    want = oracle.aggregate_packed([d.download(np.uint64, 2 * n) for d in dpk], 128 * n)
    assert np.array_equal(got, want)


def test_config2_full_size_timed_schedule(E, oracle):
    """The schedule bench.py TIMES by default ('partial-agg') at BASELINE config 2's full size: flashe_encrypt_batch_sum_dev (ten
    encrypts as one chain + the running sum of their ciphertexts, prf_chain_kernel<1024, SUM>) at n = 1e7, b = 128, C = 10 -- ALL ten
    ciphertexts against the oracle's encrypt (jzf_flashe.py:456-488), the sum against the oracle's element-wise reduce of them
    (jzf_aggregator.py:424-430), then the decrypt of that one vector against the plaintext sum (jzf_flashe.py:537-594)."""
    n, C, b, it = 10_000_000, 10, 128, 5
    eng = make(E, b)
    pts = [np.random.Generator(np.random.PCG64(1000 + c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in range(C)]
    dsum, dout = eng.alloc_vec(n), eng.alloc_vec(n)
    for d, pat in [(dsum, 0xA5), (dout, 0x3C)] + [(d, 0x5A) for d in dct]:       # nothing a skipped tile could inherit
        eng._check(eng._lib.flashe_memset_dev(eng._h, d.ptr, pat, d.nbytes))
    eng.encrypt_batch_sum_dev(it, list(range(C)), E.SCHEME_DOUBLE, n, 16, dpt, 1, dct, dsum)
    eng.decrypt_dev(it, [C], [0], n, 16, dsum, dout)
    want_sum = np.zeros((n, 2), dtype=np.uint64)
    for c in range(C):
        want = oracle.encrypt(KEY, it, c, "double", 16, b, pts[c])
        assert np.array_equal(dct[c].download(np.uint64, 2 * n).reshape(n, 2), want), c
        want_sum = oracle.aggregate_elem([want_sum, want], b)
    assert np.array_equal(dsum.download(np.uint64, 2 * n).reshape(n, 2), want_sum), "partial aggregate"
    out = dout.download(np.uint64, 2 * n).reshape(n, 2)
    lo, hi = _sum_u64(pts)
    assert np.array_equal(out[:, 0], lo) and np.array_equal(out[:, 1], hi)
    assert np.array_equal(out, oracle.decrypt(KEY, it, [C], [0], 16, b, want_sum))


def test_compact_layout_full_size_timed_schedule(E, oracle):
    """What `bench.py --bits 20 --layout u32` times: flashe_encrypt_batch_sum_u32_dev at n = 1e7, b = 20 (the width the reference's
    un-batched jobs ship), C = 10, n_jobs = 16 -- all ten uint32 ciphertexts and their sum against the oracle
    (jzf_flashe.py:19-45 slot order and chunk-dependent counters, :456-488; jzf_aggregator.py:424-430), then the decrypt of the sum."""
    n, C, b, it, J = 10_000_000, 10, 20, 9, 16
    eng = make(E, b)
    assert eng.compact_supported()
    pts = [np.random.Generator(np.random.PCG64(1500 + c)).integers(0, 2 ** b, n, dtype=np.uint64) for c in range(C)]
    d32 = [eng.upload(p.astype(np.uint32)) for p in pts]
    c32 = [eng.alloc(4 * n + 16) for _ in range(C)]
    dsum, dout = eng.alloc(4 * n + 16), eng.alloc(4 * n + 16)
    for d, pat in [(dsum, 0xA5), (dout, 0x3C)] + [(d, 0x5A) for d in c32]:
        eng._check(eng._lib.flashe_memset_dev(eng._h, d.ptr, pat, d.nbytes))
    eng.encrypt_batch_sum_u32_dev(it, list(range(C)), E.SCHEME_DOUBLE, n, J, d32, c32, dsum)
    eng.aggregate_decrypt_u32_dev(it, [C], [0], n, J, 0, n, [dsum], None, dout, 4)
    mask = np.uint64((1 << b) - 1)
    wsum = np.zeros(n, dtype=np.uint64)
    for c in range(C):
        want = oracle.encrypt(KEY, it, c, "double", J, b, pts[c])[:, 0]
        bad = np.flatnonzero(c32[c].download(np.uint32, n).astype(np.uint64) != want)
        assert bad.size == 0, (c, bad[:8])
        wsum += want
    assert np.array_equal(dsum.download(np.uint32, n).astype(np.uint64), wsum & mask), "sum of the ciphertexts"
    assert np.array_equal(dout.download(np.uint32, n).astype(np.uint64), sum(pts) & mask), "round trip"


def test_config4_size_round(E, oracle):
    """BASELINE config 4's vector (ResNet-50, n = 25 557 032, b = 128) on one GPU: round trip with a
    dropout, and the range twins reproduce the full-vector decrypt slice by slice (what the sharded
    multi-GPU round does)."""
    n, C, b, it = 25_557_032, 4, 128, 3
    eng = make(E, b)
    pts = [np.random.Generator(np.random.PCG64(3000 + c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
    dct = []
    for c in range(C):
        d_pt, d_ct = eng.upload(pts[c]), eng.alloc_vec(n)
        eng.encrypt_dev(it, c, E.SCHEME_DOUBLE, n, 16, d_pt, 1, d_ct)
        dct.append(d_ct)
        eng.sync()
        d_pt.free()
    up = [0, 1, 3]
    dagg, dout = eng.alloc_vec(n), eng.alloc_vec(n)
    eng.aggregate_elem_dev([dct[c] for c in up], n, dagg)
    add_idx, minus_idx = E.telescope(list(up))
    eng.decrypt_dev(it, add_idx, minus_idx, n, 16, dagg, dout)
    out = dout.download(np.uint64, 2 * n).reshape(n, 2)
    lo, hi = _sum_u64([pts[c] for c in up])
    assert np.array_equal(out[:, 0], lo) and np.array_equal(out[:, 1], hi)
    # slice-wise (8 "ranks", unaligned cuts) == whole
    dout2 = eng.alloc_vec(n)
    cuts = [0] + [n * k // 8 + (k * 37) % 61 for k in range(1, 8)] + [n]
    for k in range(8):
        first, count = cuts[k], cuts[k + 1] - cuts[k]
        eng.decrypt_range_dev(it, add_idx, minus_idx, n, 16, first, count, dagg.ptr + first * 16, dout2.ptr + first * 16)
    assert np.array_equal(dout2.download(np.uint64, 2 * n).reshape(n, 2), out)
    # the head of one ciphertext against the oracle
    head = 2_000_000
    ct1 = dct[1].download(np.uint64, 2 * head).reshape(head, 2)
    assert np.array_equal(ct1, oracle.encrypt(KEY, it, 1, "double", 16, b, pts[1][:head]))


@pytest.mark.parametrize("b,n,J", [(128, 20_003, 16), (100, 5000, 3), (64, 9999, 16), (23, 61_706, 16), (8, 3000, 5)])
def test_prefix_lists_of_any_length(E, oracle, b, n, J):
    """The reference sums over prefix lists of any length (jzf_flashe.py:126-150): single-mask decrypt passes one minus prefix per
    uploaded client (:311-314; BASELINE config 3 has 100 clients), and a scattered dropout pattern leaves one (add, minus) pair per
    run.  Lists longer than one launch's argument block are chained over launches that accumulate in place."""
    eng = make(E, b)
    rng = np.random.Generator(np.random.PCG64(b * 7 + n))
    ct = rand_limbs(rng, n, b)
    # single mask: 100 and 300 uploaded clients
    for C in (100, 300):
        up = list(range(C))
        got = eng.decrypt(5, [], up, J, ct)
        assert np.array_equal(got, oracle.decrypt(KEY, 5, [], up, J, b, ct)), (b, C, "single")
    # double mask: every other client of 260 dropped -> 130 runs, 130 add and 130 minus prefixes
    up = list(range(0, 260, 2))
    add_idx, minus_idx = E.telescope(list(up))
    assert len(add_idx) == 130 and len(minus_idx) == 130
    got = eng.decrypt(5, add_idx, minus_idx, J, ct)
    assert np.array_equal(got, oracle.decrypt(KEY, 5, add_idx, minus_idx, J, b, ct)), (b, "130 runs")
    # unequal list lengths, in place on the device, on a ragged range; and a long mask sum
    d = eng.upload(ct)
    first, count = 33, n - 70
    Lb = L(b)
    a_list, m_list = list(range(7, 7 + 205)), list(range(1000, 1000 + 97))
    eng.decrypt_range_dev(5, a_list, m_list, n, J, first, count, d.ptr + first * Lb * 8, d.ptr + first * Lb * 8)
    got = d.download(np.uint64, n * Lb).reshape(n, Lb)
    want = ct.copy()
    want[first:first + count] = oracle.decrypt(KEY, 5, a_list, m_list, J, b, ct)[first:first + count]
    assert np.array_equal(got, want), (b, "in place, ragged")
    assert np.array_equal(eng.mask(5, a_list, n, J), oracle.mask_sum(KEY, 5, a_list, n, J, b)), (b, "mask sum")


def test_sparse_locations_out_of_range_are_skipped_and_reported(E, oracle):
    """Device-resident location lists cannot be checked before launch: a position >= total must neither be written (the
    arbiter would corrupt HBM next to the dense vector) nor go unnoticed -- the next synchronising call fails once."""
    for b in (128, 64):
        eng = make(E, b)
        Lb = L(b)
        total, k = 5000, 40
        rng = np.random.Generator(np.random.PCG64(b))
        loc = np.sort(rng.choice(total, k, replace=False)).astype(np.uint32)
        vals = rand_limbs(rng, k, b)
        bad = loc.copy()
        bad[-1] = total + 3                                    # one entry beyond the vector
        guard = eng.alloc_vec(total + 64)                      # the dense vector plus a canary zone behind it
        canary = np.full(((total + 64), Lb), np.uint64(0x5A5A5A5A5A5A5A5A), dtype=np.uint64)
        zero = [7] + [0] * (Lb - 1)
        for sorted_lists in (False, True):
            guard.upload(canary)
            d_loc, d_vals = eng.upload(bad), eng.upload(vals)
            eng.sparse_aggregate_dev(total, [d_loc], [k], [d_vals], [zero], guard, sorted_lists=sorted_lists)
            with pytest.raises(E.FlasheError):
                eng.sync()
            eng.sync()                                         # reported once
            got = guard.download(np.uint64, (total + 64) * Lb).reshape(total + 64, Lb)
            assert np.array_equal(got[total:], canary[total:]), (b, sorted_lists, "wrote past the dense vector")
            want = oracle.expand_to_dense(total, loc[:-1], vals[:-1], np.array(zero, dtype=np.uint64), b)
            assert np.array_equal(got[:total], want), (b, sorted_lists)
        # expand_to_dense and the minus-mask scatter take the same guard
        guard.upload(canary)
        eng.expand_to_dense_dev(total, k, eng.upload(bad), eng.upload(vals), zero, guard)
        with pytest.raises(E.FlasheError):
            eng.sync()
        assert np.array_equal(guard.download(np.uint64, (total + 64) * Lb).reshape(total + 64, Lb)[total:], canary[total:])
        guard.upload(canary)
        eng.sparse_minus_mask_dev(3, [eng.upload(bad)], [k], total, 16, guard)
        with pytest.raises(E.FlasheError):
            eng.sync()
        assert np.array_equal(guard.download(np.uint64, (total + 64) * Lb).reshape(total + 64, Lb)[total:], canary[total:])
        # a good list afterwards: no stale error
        eng.sparse_minus_mask_dev(3, [eng.upload(loc)], [k], total, 16, guard)
        eng.sync()


def test_error_codes(E):
    eng = make(E, 128)
    with pytest.raises(E.FlasheError):
        eng.mask(0, [1], 10, 0)                         # n_jobs = 0
    with pytest.raises(E.FlasheError):
        E.Engine(KEY, 0)
    with pytest.raises(E.FlasheError):
        E.Engine(KEY, 128, device=99)
    with pytest.raises(E.FlasheError):
        eng.expand_to_dense(4, [9], np.zeros((1, 2), dtype=np.uint64), [0, 0])
    d = eng.alloc_vec(8)
    with pytest.raises(E.FlasheError):
        eng.aggregate_packed_dev([d], 16, 1024, d)      # out aliases an operand


@pytest.mark.parametrize("b,n,V,in_limbs", [(128, 4099, 5, 1), (128, 1000, 70, 2), (100, 33, 3, 1), (64, 5000, 9, 1), (20, 777, 130, 1), (128, 0, 3, 1), (64, 10, 0, 1)])
def test_combine_batch_sum_vs_oracle(E, oracle, b, n, V, in_limbs):
    """flashe_combine_batch_sum_dev: out[v] = in[v] + add[v] - minus[v] (jzf_flashe.py:480-481 with precomputed masks) for V vectors and
    sum_out = the element-wise reduce of the results (jzf_aggregator.py:424-430) from the same pass -- more vectors than one launch's
    table holds (running sum across launches), missing add / minus entries, two-limb inputs, empty shapes, misuse."""
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(b * 1000 + n + V))
    hi = 2 ** min(b, 64)
    def vec(limbs):
        a = np.zeros((n, limbs), dtype=np.uint64)
        a[:, 0] = rng.integers(0, hi, n, dtype=np.uint64) if hi < 2 ** 64 else rng.integers(0, 2 ** 64, n, dtype=np.uint64)
        if limbs == 2:
            a[:, 1] = rng.integers(0, 2 ** (b - 64), n, dtype=np.uint64)
        return a
    ins = [vec(in_limbs) for _ in range(V)]
    adds = [vec(Lb) if v % 3 != 2 else None for v in range(V)]
    mins = [vec(Lb) if v % 4 != 1 else None for v in range(V)]
    d_in = [eng.upload(x) for x in ins]
    d_add = [eng.upload(x) if x is not None else None for x in adds]
    d_min = [eng.upload(x) if x is not None else None for x in mins]
    outs = [eng.alloc_vec(max(n, 1)) for _ in range(V)]
    dsum = eng.alloc_vec(max(n, 1))
    eng._check(eng._lib.flashe_memset_dev(eng._h, dsum.ptr, 0xC3, dsum.nbytes))
    eng.combine_batch_sum_dev(n, d_in, in_limbs, d_add, d_min, outs, dsum)
    want = [oracle.combine(b, ins[v], adds[v], mins[v]) for v in range(V)]
    for v in range(V):
        assert np.array_equal(outs[v].download(np.uint64, n * Lb).reshape(n, Lb), want[v]), (b, n, v)
    if n:
        wsum = oracle.aggregate_elem(want, b) if V else np.zeros((n, Lb), dtype=np.uint64)
        assert np.array_equal(dsum.download(np.uint64, n * Lb).reshape(n, Lb), wsum), (b, n, V, "sum")
    if n and V:
        with pytest.raises(E.FlasheError):
            eng.combine_batch_sum_dev(n, d_in, in_limbs, d_add, d_min, outs, outs[0])          # the sum must not alias an output
        if in_limbs == Lb:
            with pytest.raises(E.FlasheError):
                eng.combine_batch_sum_dev(n, d_in, in_limbs, d_add, d_min, outs, d_in[V - 1])   # ... nor an operand


@pytest.mark.parametrize("b,n,V,in_limbs,minus", [(128, 4099, 5, 1, True), (128, 1000, 70, 2, False), (128, 61_706, 100, 1, False), (128, 900, 125, 1, False),
                                                  (100, 33, 3, 1, True), (64, 5000, 9, 1, False), (20, 777, 130, 1, True), (23, 61_706, 100, 1, False),
                                                  (128, 0, 3, 1, False), (64, 10, 0, 1, False)])
def test_combine_batch_sum_decrypt_vs_oracle(E, oracle, b, n, V, in_limbs, minus):
    """flashe_combine_batch_sum_decrypt_dev (round 6): the online encrypts out[v] = in[v] + add[v] - minus[v] with precomputed masks
    (jzf_flashe.py:457, :480-481), their element-wise reduce (jzf_aggregator.py:424-430) AND the decrypt of that reduce with precomputed
    decrypt masks (jzf_flashe.py:557-571) from one pass.  Batches without a minus operand take the three-pointer table (up to 120
    vectors per launch: config 3's hundred clients in one; 125 / 130 carry the sum across launches and decrypt in the last one), batches
    with one the four-pointer table of 64; every output, the sum and the decrypted vector against the oracle; misuse is refused."""
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(b * 999 + n + V))
    hi = 2 ** min(b, 64)

    def vec(limbs):
        a = np.zeros((n, limbs), dtype=np.uint64)
        a[:, 0] = rng.integers(0, hi, n, dtype=np.uint64) if hi < 2 ** 64 else rng.integers(0, 2 ** 64, n, dtype=np.uint64)
        if limbs == 2:
            a[:, 1] = rng.integers(0, 2 ** (b - 64), n, dtype=np.uint64)
        return a
    ins = [vec(in_limbs) for _ in range(V)]
    adds = [vec(Lb) if v % 5 != 4 else None for v in range(V)]
    mins = [vec(Lb) if (minus and v % 4 != 1) else None for v in range(V)]
    dec_add, dec_min = vec(Lb), (vec(Lb) if V % 2 else None)
    d_in = [eng.upload(x) for x in ins]
    d_add = [eng.upload(x) if x is not None else None for x in adds]
    d_min = [eng.upload(x) if x is not None else None for x in mins] if minus else None
    outs = [eng.alloc_vec(max(n, 1)) for _ in range(V)]
    dsum, ddec = eng.alloc_vec(max(n, 1)), eng.alloc_vec(max(n, 1))
    d_da, d_dm = eng.upload(dec_add) if n else eng.alloc_vec(1), (eng.upload(dec_min) if (dec_min is not None and n) else None)
    for buf, pat in [(dsum, 0xC3), (ddec, 0x3C)] + [(o, 0x77) for o in outs]:
        eng.memset_dev(buf, pat, buf.nbytes)
    eng.combine_batch_sum_decrypt_dev(n, d_in, in_limbs, d_add, d_min, outs, dsum, d_da, d_dm, ddec)
    want = [oracle.combine(b, ins[v], adds[v], mins[v]) for v in range(V)]
    for v in range(V):
        assert np.array_equal(outs[v].download(np.uint64, n * Lb).reshape(n, Lb), want[v]), (b, n, v)
    if n:
        wsum = oracle.aggregate_elem(want, b) if V else np.zeros((n, Lb), dtype=np.uint64)
        assert np.array_equal(dsum.download(np.uint64, n * Lb).reshape(n, Lb), wsum), (b, n, V, "sum")
        wdec = oracle.combine(b, wsum, dec_add, dec_min if d_dm is not None else None)
        assert np.array_equal(ddec.download(np.uint64, n * Lb).reshape(n, Lb), wdec), (b, n, V, "decrypt of the sum")
    if n and V:
        for bad in (dict(dec_out=dsum), dict(dec_out=outs[0]), dict(dec_out=d_in[0]) if in_limbs == Lb else dict(dec_out=dsum), dict(sum_out=outs[-1]), dict(dec_add=dsum)):
            kw = dict(sum_out=dsum, dec_add=d_da, dec_out=ddec)
            kw.update(bad)
            with pytest.raises(E.FlasheError):
                eng.combine_batch_sum_decrypt_dev(n, d_in, in_limbs, d_add, d_min, outs, kw["sum_out"], kw["dec_add"], d_dm, kw["dec_out"])


def test_double_mask_idx_range_at_the_raw_abi(E, oracle):
    """jzf_flashe.py:352-353: the double mask's minus prefix is (self.idx + 1).to_bytes(4, 'big') -- OverflowError for idx = 2^32 - 1.
    The raw C ABI refuses the same value (FLASHE_EINVAL) instead of wrapping to prefix 0, in every encrypt entry point; the single mask
    (no idx + 1) and idx = 2^32 - 2 still work and equal the oracle."""
    top = 0xFFFFFFFF
    for b in (128, 20):
        eng = make(E, b)
        n = 64
        pt = np.arange(n, dtype=np.uint64)
        dp, dc, ds = eng.upload(pt), eng.alloc_vec(n), eng.alloc_vec(n)
        bad = [lambda: eng.encrypt(3, top, E.SCHEME_DOUBLE, 4, pt),
               lambda: eng.encrypt_dev(3, top, E.SCHEME_DOUBLE, n, 4, dp, 1, dc),
               lambda: eng.encrypt_range_dev(3, top, E.SCHEME_DOUBLE, n, 4, 0, n, dp, 1, dc),
               lambda: eng.encrypt_batch_dev(3, [0, top], E.SCHEME_DOUBLE, n, 4, [dp, dp], 1, [dc, ds]),
               lambda: eng.encrypt_batch_sum_dev(3, [top], E.SCHEME_DOUBLE, n, 4, [dp], 1, [dc], ds),
               lambda: eng.encrypt_batch_range_dev(3, [top], E.SCHEME_DOUBLE, n, 4, 0, n, [dp], 1, [dc]),
               lambda: eng.prepare_encrypt(4, top, E.SCHEME_DOUBLE, n, 4)]
        if b <= 32:
            p32, c32 = eng.upload(pt.astype(np.uint32)), eng.alloc(4 * n)
            bad.append(lambda: eng.encrypt_batch_u32_dev(3, [top], E.SCHEME_DOUBLE, n, 4, [p32], [c32]))
        x = eng.upload(np.linspace(-1, 1, n).astype(np.float32))
        u = eng.upload(np.full(n, 0.5))
        bad.append(lambda: eng.quantize_encrypt_dev(3, top, E.SCHEME_DOUBLE, n, 4, x, False, 2.0, 16, u, dc))
        bad.append(lambda: eng.quantize_encrypt_model_dev(3, top, E.SCHEME_DOUBLE, n, 4, 0, n, [(0, x.ptr, 2.0, False)], 16, u, dc))
        for k, call in enumerate(bad):
            with pytest.raises(E.FlasheError) as ei:
                call()
            assert ei.value.code == -22 and "2^32" in str(ei.value), (b, k, str(ei.value))
        # neighbours of the refused value
        assert np.array_equal(eng.encrypt(3, top, E.SCHEME_SINGLE, 4, pt), oracle.encrypt(KEY, 3, top, "single", 4, b, pt))
        assert np.array_equal(eng.encrypt(3, top - 1, E.SCHEME_DOUBLE, 4, pt), oracle.encrypt(KEY, 3, top - 1, "double", 4, b, pt))
        assert eng.compact_supported() == (b <= 32)


@pytest.mark.parametrize("b,n", [(128, 1_300_003), (20, 2_500_001), (64, 2_100_000)])
def test_pipelined_host_twins_with_pinned_results(E, oracle, b, n, monkeypatch):
    """The host-pointer twins cut large vectors into chunks (upload + kernel of chunk q beside the download of chunk q - 1) when the
    result array is page-locked -- the engine's result pool with FLASHE_HOST_POOL_PINNED=1.  Same bytes as the one-shot form and the
    oracle: encrypt (double / single), a dropout decrypt with prefix lists, the reduce; ragged last chunk, chunk size shrunk to 4 MB so
    that several chunks and both hand-off events take part; b <= 64 chunk boundaries fall inside the reference's n_jobs chunks."""
    monkeypatch.setenv("FLASHE_HOST_POOL_PINNED", "1")
    monkeypatch.setenv("FLASHE_TWIN_CHUNK_MB", "4")
    monkeypatch.setattr(E, "_HOST_POOL", E._HostPool())
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(n))
    pts = [rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for _ in range(3)]
    cts = [eng.encrypt(5, c, E.SCHEME_DOUBLE, 7, pts[c]) for c in range(3)]
    for c in range(3):
        assert np.array_equal(cts[c], oracle.encrypt(KEY, 5, c, "double", 7, b, pts[c])), (b, c)
    assert np.array_equal(eng.encrypt(5, 9, E.SCHEME_SINGLE, 7, pts[0]), oracle.encrypt(KEY, 5, 9, "single", 7, b, pts[0]))
    agg = eng.aggregate_elem(cts)
    assert np.array_equal(agg, oracle.aggregate_elem(cts, b))
    dec = eng.decrypt(5, [3, 7], [0, 5], 7, agg)
    assert np.array_equal(dec, oracle.decrypt(KEY, 5, [3, 7], [0, 5], 7, b, agg))
    # the pageable-result form of the same calls (no pipeline) gives the same arrays
    monkeypatch.setenv("FLASHE_TWIN_PIPELINE", "0")
    assert np.array_equal(eng.encrypt(5, 1, E.SCHEME_DOUBLE, 7, pts[1]), cts[1]) and np.array_equal(eng.aggregate_elem(cts), agg)


def test_device_block_cache_reuse_is_safe_across_streams(E, oracle):
    """flashe_dev_free parks a block, flashe_dev_alloc hands it out again (no hipMalloc / hipFree pair per DeviceVector) -- but only
    after a device-wide synchronisation, so a kernel of ANOTHER ctx that still reads the block when its owner drops it sees the old
    contents to the end: a 32-operand reduce on ctx B reads a vector that ctx A frees, re-allocates (same address) and overwrites."""
    import ctypes
    a, b_ = make(E, 128), make(E, 128)
    n = 3_000_001
    rng = np.random.Generator(np.random.PCG64(4))
    x = rng.integers(0, 2 ** 64, size=(n, 2), dtype=np.uint64)
    want = oracle.aggregate_elem([x] * 32, 128)
    dx = a.upload(x)
    lib = a._lib
    h0, m0 = ctypes.c_uint64(), ctypes.c_uint64()
    lib.flashe_dev_pool_stats(0, None, ctypes.byref(h0), ctypes.byref(m0))
    for rep in range(4):
        d = a.alloc_vec(n)
        a.combine_dev(n, dx, 2, None, None, d)                      # d <- x on A's stream
        a.sync()
        out = b_.alloc_vec(n)
        b_.aggregate_elem_dev([d] * 32, n, out)                     # B reads d for a while (1.5 GB of loads)
        ptr = d.ptr
        d.free()                                                    # parked while B still reads it
        d2 = a.alloc_vec(n)
        assert d2.ptr == ptr, "the parked block was not reused"
        a._check(lib.flashe_memset_dev(a._h, d2.ptr, 0xFF, d2.nbytes))
        a.sync()
        assert np.array_equal(out.download(np.uint64, 2 * n).reshape(n, 2), want), rep
        d2.free()
        out.free()
    h1, m1, parked = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    lib.flashe_dev_pool_stats(0, ctypes.byref(parked), ctypes.byref(h1), ctypes.byref(m1))
    assert h1.value - h0.value >= 7 and parked.value >= 2 * (48 << 20)
    assert lib.flashe_dev_trim(0) == 0
    lib.flashe_dev_pool_stats(0, ctypes.byref(parked), None, None)
    assert parked.value == 0
    d3 = a.alloc_vec(n)                                             # still works after a trim
    a.combine_dev(n, dx, 2, None, None, d3)
    assert np.array_equal(d3.download(np.uint64, 2 * n).reshape(n, 2), x)


@pytest.mark.parametrize("b,total,C,frac", [(128, 20_011, 5, 0.2), (64, 30_001, 4, 0.3), (20, 9_999, 7, 0.5), (128, 300_007, 70, 0.02), (7, 5000, 3, 0.9),
                                             (128, 1000, 1, 0.1), (100, 4096, 3, 1.0)])
def test_sparse_double_masks_from_location_lists(E, oracle, b, total, C, frac):
    """flashe_sparse_double_masks_dev (the sparse branch of set_idx_list for the double mask: run analysis + dense-position masks,
    jzf_flashe.py:388-426, :155-225) straight from the clients' sorted location lists, against the reference's own formulation --
    one-hot vectors, per-position run analysis on the host, _static_prepare_decrypt_spar through the oracle -- and against the
    selector-based entry point.  Neighbouring clients share many positions (their masks cancel there), one list is empty, one full;
    more clients than one launch group holds (70 > 64)."""
    eng = make(E, b)
    Lb = L(b)
    rng = np.random.Generator(np.random.PCG64(total + C))
    base = np.sort(rng.choice(total, max(1, int(total * frac)), replace=False))
    locs = []
    for c in range(C):
        keep = base[rng.random(len(base)) < 0.7]                               # heavy overlap between neighbours
        extra = rng.choice(total, max(1, len(base) // 10), replace=False)
        locs.append(np.unique(np.concatenate([keep, extra])).astype(np.uint32))
    if C >= 4:
        locs[2] = np.zeros(0, dtype=np.uint32)                                 # a client that uploaded nothing
    if frac == 1.0:
        locs[1] = np.arange(total, dtype=np.uint32)
    ohs = []
    for l in locs:
        a = np.zeros(total, dtype=np.uint8)
        a[l] = 1
        ohs.append(a)
    minus = [ohs[c] & (1 - ohs[c - 1]) if c > 0 else ohs[c] for c in range(C)]
    add = [np.zeros(total, dtype=np.uint8)] + [ohs[c] & (1 - ohs[c + 1]) if c < C - 1 else ohs[c] for c in range(C)]
    want_add, want_minus = oracle.sparse_dense_mask(KEY, 11, add, total, b), oracle.sparse_dense_mask(KEY, 11, minus, total, b)
    dloc = [eng.upload(l) if len(l) else eng.alloc(16) for l in locs]
    da, dm = eng.alloc_vec(total), eng.alloc_vec(total)
    eng._check(eng._lib.flashe_memset_dev(eng._h, da.ptr, 0x5A, da.nbytes))
    eng.sparse_double_masks_dev(11, dloc, [len(l) for l in locs], total, da, dm)
    assert np.array_equal(da.download(np.uint64, total * Lb).reshape(total, Lb), want_add), (b, total, "add")
    assert np.array_equal(dm.download(np.uint64, total * Lb).reshape(total, Lb), want_minus), (b, total, "minus")
    if C <= 7:
        assert np.array_equal(eng.sparse_dense_mask(11, add, total), want_add)        # the selector-based entry point agrees
    # a list that is not strictly increasing is reported, nothing is written out of bounds
    if C >= 2 and len(locs[0]) > 3:
        bad = locs[0].copy()
        bad[1], bad[2] = bad[2], bad[1]
        dloc[0] = eng.upload(bad)
        eng.sparse_double_masks_dev(11, dloc, [len(l) for l in locs], total, da, dm)
        with pytest.raises(E.FlasheError):
            eng.sync()
            da.download(np.uint64, 2)


@pytest.mark.parametrize("b,scheme", [(128, "double"), (128, "single"), (20, "double"), (64, "double")])
def test_device_resident_encrypts_reuse_the_ciphers_upload_block(oracle, b, scheme):
    """Round 6: FlasheCipher.encrypt(host array, device=True) stages the plaintext in the cipher's OWN upload block
    (no fresh block per call: the caching allocator would synchronise the device before it hands the previous call's block out again,
    which serialised the clients' uploads and encrypts) -- vectors of changing length through the same cipher (the block grows, is
    kept, shrinks), two ciphers interleaved, uint64 / uint32 / object inputs: every ciphertext equals the oracle's
    (jzf_flashe.py:431-488), and a handle returned earlier still holds ITS ciphertext after later calls reused the block."""
    from flashe_amd import cipher as cm
    cm.N_JOBS = 16
    it = 11
    rng = np.random.Generator(np.random.PCG64(b + len(scheme)))
    ciphers = []
    for c in range(2):
        ci = cm.FlasheCipher(b, mask=scheme)
        ci.set_num_clients(2)
        ci.generate_prp_seed(KEY)
        ci.set_iter_index(it)
        ci.idx = c
        ciphers.append(ci)
    kept = []
    for n in (100_003, 40_001, 100_003, 250_000, 17, 250_000, 3_000):
        for c, ci in enumerate(ciphers):
            pt = rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64)
            want = oracle.encrypt(KEY, it, c, scheme, 16, b, pt)
            form = pt.astype(np.uint32) if (b <= 32 and n % 2) else (pt.astype(object) if n == 17 else pt)
            h = ci.encrypt(form, device=True)
            got = h.to_host()
            got = got.astype(np.uint64).reshape(n, -1)
            assert np.array_equal(got, want[:, :got.shape[1]]), (b, scheme, n, c)
            assert h.buf is not ci._pt_stage and h.buf.ptr != ci._pt_stage.ptr          # the result never lives in the upload block
            assert ci._pt_stage.nbytes >= pt.nbytes // (2 if (b <= 32 and n % 2) else 1) or b <= 32
            kept.append((h, want))
    for h, want in kept:                                                             # earlier results untouched by later uploads
        got = h.to_host().astype(np.uint64).reshape(len(h), -1)
        assert np.array_equal(got, want[:, :got.shape[1]])
    blocks = {ci._pt_stage.ptr for ci in ciphers}
    assert len(blocks) == 2
    ciphers[0].release_device_buffers()
    assert ciphers[0]._pt_stage is None
    h = ciphers[0].encrypt(np.arange(5, dtype=np.uint64), device=True)               # ... and comes back on demand
    assert np.array_equal(h.to_host().astype(np.uint64).reshape(5, -1)[:, 0], oracle.encrypt(KEY, it, 0, scheme, 16, b, np.arange(5, dtype=np.uint64))[:, 0])


@pytest.mark.parametrize("b,scheme", [(20, "double"), (32, "double"), (8, "single"), (23, "double")])
def test_flashe_cipher_compact_layout(oracle, b, scheme):
    """with int_bits <= 32 the drop-in class keeps device-resident vectors as uint32 arrays (half the bytes of the one-limb
    layout) and takes / returns np.uint32 arrays: encrypt (flashe_encrypt_batch_u32_dev), aggregate (flashe_aggregate_elem_u32_dev),
    the no-dropout decrypt (flashe_aggregate_decrypt_u32_dev on the vector itself), a dropout decrypt (prefix lists: widened on the
    device and narrowed back) and the packed reduce -- every value against the oracle, handles and host arrays, mixed operands."""
    from flashe_amd import cipher as cm
    cm.N_JOBS = 16
    n, C, it = 70_001, 4, 5
    rng = np.random.Generator(np.random.PCG64(b))
    pts = [rng.integers(0, 2 ** min(b, 16), n, dtype=np.uint64) for _ in range(C)]
    ciphers = []
    for c in range(C):
        ci = cm.FlasheCipher(b, mask=scheme)
        ci.set_num_clients(C)
        ci.generate_prp_seed(KEY)
        ci.set_iter_index(it)
        ci.idx = c
        ciphers.append(ci)
    want_ct = [oracle.encrypt(KEY, it, c, scheme, 16, b, pts[c]) for c in range(C)]
    hs = []
    for c in range(C):
        host32 = ciphers[c].encrypt(pts[c].astype(np.uint32))                        # uint32 in -> uint32 out
        assert host32.dtype == np.uint32 and np.array_equal(host32.astype(np.uint64), want_ct[c][:, 0])
        h = ciphers[c].encrypt(pts[c].astype(object) if c % 2 else pts[c], device=True)   # object ints / uint64 in -> compact handle
        assert h.compact and h.to_host().dtype == np.uint32 and np.array_equal(h.to_host().astype(np.uint64), want_ct[c][:, 0])
        h2 = ciphers[c].encrypt(h)                                                    # a compact handle as plaintext
        assert h2.compact
        hs.append(h)
    want_agg = oracle.aggregate_elem(want_ct, b)
    agg = ciphers[0].aggregate(hs)
    assert agg.compact and np.array_equal(agg.to_host().astype(np.uint64), want_agg[:, 0])
    agg_host = ciphers[0].aggregate([h.to_host() for h in hs])                        # uint32 arrays in -> uint32 array out
    assert agg_host.dtype == np.uint32 and np.array_equal(agg_host, agg.to_host())
    mixed = ciphers[0].aggregate([hs[0], want_ct[1]] + hs[2:], device=False)          # a one-limb host operand among compact handles
    assert np.array_equal(np.asarray(mixed).reshape(n, -1)[:, 0].astype(np.uint64), want_agg[:, 0])
    aggp = ciphers[0].aggregate(hs, packed=True)
    want_aggp = oracle.unpack(oracle.aggregate_packed([oracle.pack(ct, b) for ct in want_ct], n * b), n, b)
    assert np.array_equal(aggp.to_host().reshape(n, -1)[:, 0], want_aggp[:, 0])
    # decrypt: everybody uploaded (one add / one minus prefix for the double mask), then client 1 dropped (prefix lists)
    for up in (list(range(C)), [0, 2, 3]):
        a = ciphers[0].aggregate([hs[c] for c in up])
        if scheme == "double":
            add, minus = cm._engine.telescope(sorted(up))
        else:
            add, minus = [], sorted(up)
        want = oracle.decrypt(KEY, it, add, minus, 16, b, oracle.aggregate_elem([want_ct[c] for c in up], b))
        ciphers[0].set_idx_list(raw_idx_list=list(up), mode="decrypt")
        dec = ciphers[0].decrypt(a)
        assert dec.compact and np.array_equal(dec.to_host().astype(np.uint64), want[:, 0]), (b, scheme, up)
        ciphers[0].set_idx_list(raw_idx_list=list(up), mode="decrypt")
        dec_host = ciphers[0].decrypt(a.to_host())                                    # uint32 array in -> uint32 array out
        assert dec_host.dtype == np.uint32 and np.array_equal(dec_host.astype(np.uint64), want[:, 0])
        ciphers[0].set_idx_list(raw_idx_list=list(up), mode="decrypt")
        dec_obj = ciphers[0].decrypt(a, device=False)
        assert dec_obj.dtype == np.uint32
    assert np.array_equal(want[:, 0], sum(pts[c] for c in up) & np.uint64((1 << b) - 1))


@pytest.mark.parametrize("b", [128, 64, 20])
def test_arbiter_reduce_runs_unchanged_on_device_handles(oracle, b):
    """VERDICT r3 missing #4: the arbiter's literal reduce -- reduce(lambda x, y: (x + y) % mod, models) with mod = 1 << int_bits
    (jzf_aggregator.py:424-430) over weights objects whose `+` and `%` map over the layers (JZFOrderDictWeights.binary_op / map_values,
    jzf_weights.py:340-341, :355-357, :446-472) -- runs as written when the layers are DeviceVector handles: `+` is the element-wise
    mod-add on the device, `% (1 << int_bits)` the identity it is for reduced vectors."""
    import operator
    from functools import reduce
    from flashe_amd import cipher as cm

    class Weights:                                     # the two operators of JZFOrderDictWeights the reduce uses, as the reference wrote them
        def __init__(self, d):
            self._weights, self.walking_order = d, sorted(d.keys(), key=str)

        def __add__(self, other):                      # binary_op(other, operator.add, inplace=False), jzf_weights.py:355-357, :460-472
            return Weights({k: operator.add(other._weights[k], self._weights[k]) for k in self.walking_order})

        def __mod__(self, other):                      # map_values(lambda x: x % other, inplace=False), :340-341
            return Weights({k: self._weights[k] % other for k in self.walking_order})

    cm.N_JOBS = 16
    n, C, it = 50_003, 5, 2
    rng = np.random.Generator(np.random.PCG64(b))
    models, cts = [], []
    for c in range(C):
        ci = cm.FlasheCipher(b)
        ci.set_num_clients(C)
        ci.generate_prp_seed(KEY)
        ci.set_iter_index(it)
        ci.idx = c
        pt = rng.integers(0, 2 ** min(b, 16), n, dtype=np.uint64)
        models.append(Weights({"flat": ci.encrypt(pt, device=True)}))
        cts.append(oracle.encrypt(KEY, it, c, "double", 16, b, pt))
    mod = 1 << b
    agg = reduce(lambda x, y: (x + y) % mod, models)
    got = agg._weights["flat"]
    assert isinstance(got, cm.DeviceVector)
    want = oracle.aggregate_elem(cts, b)
    assert np.array_equal(np.asarray(got.to_host()).reshape(n, -1).astype(np.uint64), want)
    assert sum(m._weights["flat"] for m in models) is not None            # sum() starts from 0
    with pytest.raises(ValueError):
        _ = got % (mod + 2)


def test_span_bounds_handle_shared_by_aggregate_and_decrypt(E, oracle):
    """flashe_span_bounds_*: the span bounds of a round's location lists computed once and handed to the sparse aggregate AND the sparse
    decrypt -- the same dense vectors as when each pass computes them itself; more clients than one group holds; recompute for the next
    round's lists; a handle used with other lists is refused."""
    from flashe_amd._lib import FlasheError
    eng = make(E, 128)
    rng = np.random.Generator(np.random.PCG64(4))
    for C, total, k in [(5, 100_003, 2_000), (70, 50_000, 300), (1, 9_000, 9_000)]:
        locs = [np.sort(rng.choice(total, k, replace=False)).astype(np.uint32) for _ in range(C)]
        vals = [rng.integers(0, 2 ** 63, (k, 2), dtype=np.uint64) for _ in range(C)]
        dl, dv = [eng.upload(l) for l in locs], [eng.upload(v) for v in vals]
        zeros = [7 + c for c in range(C)]
        a0, a1, d0, d1 = (eng.alloc_vec(total) for _ in range(4))
        eng.sparse_aggregate_dev(total, dl, [k] * C, dv, zeros, a0, sorted_lists=True)
        eng.sparse_decrypt_dev(3, dl, [k] * C, total, 16, a0, d0, sorted_lists=True)
        bnd = eng.span_bounds(total, dl, [k] * C)
        eng.sparse_aggregate_dev(total, dl, [k] * C, dv, zeros, a1, bounds=bnd)
        eng.sparse_decrypt_dev(3, dl, [k] * C, total, 16, a1, d1, bounds=bnd)
        assert np.array_equal(a0.download(np.uint64, 2 * total), a1.download(np.uint64, 2 * total)), (C, total, k, "aggregate")
        assert np.array_equal(d0.download(np.uint64, 2 * total), d1.download(np.uint64, 2 * total)), (C, total, k, "decrypt")
        want = oracle.sparse_minus_mask(KEY, 3, locs, total, 16, 128)
        agg = a0.download(np.uint64, 2 * total).reshape(total, 2)
        assert np.array_equal(d1.download(np.uint64, 2 * total).reshape(total, 2), oracle.combine(128, agg, None, want))
        # next round: other lists, the handle recomputed in place
        locs2 = [np.sort(rng.choice(total, k, replace=False)).astype(np.uint32) for _ in range(C)]
        dl2 = [eng.upload(l) for l in locs2]
        with pytest.raises(FlasheError):
            eng.sparse_aggregate_dev(total, dl2, [k] * C, dv, zeros, a1, bounds=bnd)       # bounds of OTHER lists: refused
        bnd.recompute(dl2, [k] * C)
        eng.sparse_aggregate_dev(total, dl2, [k] * C, dv, zeros, a1, bounds=bnd)
        eng.sparse_aggregate_dev(total, dl2, [k] * C, dv, zeros, a0, sorted_lists=True)
        assert np.array_equal(a0.download(np.uint64, 2 * total), a1.download(np.uint64, 2 * total)), (C, "recomputed")


@pytest.mark.parametrize("b,C,total,k,pt_limbs", [(128, 5, 100_003, 2_000, 1), (128, 70, 50_000, 300, 2), (100, 9, 30_011, 1_234, 1), (128, 1, 9_000, 9_000, 2),
                                                  (128, 3, 1_759, 40, 1), (128, 3, 1_761, 1_761, 2), (64, 4, 20_000, 700, 1), (20, 3, 20_000, 700, 1)])
def test_sparse_encrypt_aggregate_in_one_pass(E, oracle, b, C, total, k, pt_limbs):
    """flashe_sparse_encrypt_aggregate_dev (the span reduce with the PRF inside): every client's compact ciphertext against the oracle's
    single-mask encrypt, the aggregate against the sum of the oracle's expand_to_dense of those ciphertexts; client indices that are not
    0 .. C-1, clients with ragged list lengths and an empty one, more clients than one group holds, a vector a few positions around one
    span, with and without the bounds handle; int_bits <= 64 takes the two calls it stands for."""
    eng = make(E, b)
    L = 2 if b > 64 else 1
    if L == 1 and pt_limbs == 2:
        pytest.skip("one-limb ctx")
    rng = np.random.Generator(np.random.PCG64(b * 1000 + C))
    ks = [k] * C
    if C >= 3:
        ks[1] = max(k // 3, 1)
        ks[2] = 0
    idx = [(3 * c + 1) % 97 for c in range(C)]
    locs = [np.sort(rng.choice(total, kc, replace=False)).astype(np.uint32) for kc in ks]
    top = 2 ** 63 if b >= 64 else 2 ** (b - 2)
    pts = [rng.integers(0, top, (kc, pt_limbs), dtype=np.uint64) for kc in ks]
    if b < 128 and pt_limbs == 2:
        for p in pts:
            p[:, 1] &= np.uint64((1 << (b - 64)) - 1)
    zeros = [(1 << min(b - 4, 20)) + c for c in range(C)]
    dl = [eng.upload(l) if l.size else eng.alloc(16) for l in locs]
    dp = [eng.upload(p) if p.size else eng.alloc(16) for p in pts]
    want_ct = []
    for c in range(C):
        p = pts[c] if pt_limbs == L else np.concatenate([pts[c], np.zeros((ks[c], 1), dtype=np.uint64)], axis=1)
        want_ct.append(oracle.encrypt(KEY, 5, idx[c], "single", 16, b, p) if ks[c] else np.zeros((0, L), dtype=np.uint64))
    want = np.zeros((total, L), dtype=np.uint64)
    for c in range(C):
        z = np.array([[zeros[c]] + [0] * (L - 1)], dtype=np.uint64)
        want = oracle.aggregate_elem([want, oracle.expand_to_dense(total, locs[c], want_ct[c], z, b)], b)
    for with_bounds in (False, True):
        bnd = eng.span_bounds(total, dl, ks) if with_bounds else None
        cts = [eng.alloc_vec(max(kc, 1)) for kc in ks]
        agg = eng.alloc_vec(total)
        eng.sparse_encrypt_aggregate_dev(5, idx, dl, ks, dp, pt_limbs, zeros, total, 16, cts, agg, bounds=bnd)
        for c in range(C):
            if ks[c]:
                assert np.array_equal(cts[c].download(np.uint64, ks[c] * L).reshape(ks[c], L), want_ct[c]), (b, C, c, with_bounds, "ciphertext")
        assert np.array_equal(agg.download(np.uint64, total * L).reshape(total, L), want), (b, C, with_bounds, "aggregate")


def test_span_prf_crowded_dense_sparse_and_empty_spans(E, oracle):
    """The span passes with the PRF inside under stress (the barrier-free phase counters; round 6: the per-window round-2
    table): ~1,500 spans, six per workgroup, whose density changes from span to span -- crowded (more than 1,024 entries: second pass,
    tables published between barriers), one client holding EVERY position (more than 256 entries of a client per span: entries beyond
    its second counter window are computed in full), the usual 1 %, empty, and a client whose slice straddles a 256-entry window in
    almost every span.  Several rounds back to back (a late wave of one launch meets the next launch's tables); every ciphertext, the
    aggregate and the decrypted vector against the oracle (jzf_flashe.py:316-343, jzf_aggregator.py:150-165, :424-430)."""
    b, J, C = 128, 16, 6
    eng = make(E, b)
    span = eng.sparse_span()
    n_spans = 6 * 256 + 3
    total = span * n_spans - 7
    rng = np.random.Generator(np.random.PCG64(606))
    mode = np.arange(n_spans) % 5
    # per (mode, client) density of a span
    dens = np.array([[0.30, 0.30, 0.25, 0.02, 0.0, 0.01],       # crowded: ~1,450 entries
                     [0.01, 0.01, 0.01, 0.01, 0.01, 0.01],      # the usual
                     [0.0, 0.0, 0.0, 0.0, 0.0, 0.0],            # empty
                     [1.0, 0.01, 0.0, 0.2, 0.0, 0.01],          # one client holds every position of the span, another a fifth
                     [0.05, 0.0, 0.0, 0.0, 0.6, 0.0]])          # 1,000 entries of one client (four windows), 90 of another
    p_of_pos = dens[np.repeat(mode, span)[:total]]               # [total, C]
    locs = [np.flatnonzero(rng.random(total) < p_of_pos[:, c]).astype(np.uint32) for c in range(C)]
    ks = [int(l.size) for l in locs]
    assert max(ks) > 300_000 and min(ks) > 10_000
    idx = [7, 8, 30, 2, 0, 55]
    zeros = [1 << 31, 5, (1 << 40) + 3, 0, 77, 1 << 20]
    dl = [eng.upload(l) for l in locs]
    agg, dec = eng.alloc_vec(total), eng.alloc_vec(total)
    bnd = eng.span_bounds(total, dl, ks)
    for it in (3, 4, 5):
        pts = [rng.integers(0, 2 ** 64, kc, dtype=np.uint64) for kc in ks]
        dp = [eng.upload(p) for p in pts]
        cts = [eng.alloc_vec(kc) for kc in ks]
        for buf in cts + [agg, dec]:
            eng.memset_dev(buf, 0xd7, buf.nbytes)
        # two rounds in a row on the same stream: the second launch's prologue overlaps the first one's last spans
        eng.sparse_encrypt_aggregate_dev(it, idx, dl, ks, dp, 1, zeros, total, J, cts, agg, bounds=bnd)
        eng.sparse_decrypt_dev(it, dl, ks, total, J, agg, dec, sorted_lists=True, bounds=bnd)
        want = np.zeros((total, 2), dtype=np.uint64)
        for c in range(C):
            want_ct = oracle.encrypt(KEY, it, idx[c], "single", J, b, pts[c])
            assert np.array_equal(cts[c].download(np.uint64, 2 * ks[c]).reshape(ks[c], 2), want_ct), (it, c, "ciphertext")
            z = np.array([[zeros[c], 0]], dtype=np.uint64)
            want = oracle.aggregate_elem([want, oracle.expand_to_dense(total, locs[c], want_ct, z, b)], b)
        assert np.array_equal(agg.download(np.uint64, 2 * total).reshape(total, 2), want), (it, "aggregate")
        # the decrypt twin subtracts term(iter, c, q) of client NUMBER c = 0 .. C-1 (the decrypting party's view of the uploads)
        mask = oracle.sparse_minus_mask(KEY, it, locs, total, J, b)
        assert np.array_equal(dec.download(np.uint64, 2 * total).reshape(total, 2), oracle.combine(b, want, None, mask)), (it, "decrypt")


def test_span_bounds_with_lists_at_any_alignment(E, oracle):
    """The bounds pass reads 16 bytes per load when every list is 16-byte aligned and entry by entry otherwise: lists that start 4 / 8 / 12
    bytes into a buffer give the same sparse aggregate and decrypt as aligned ones (list lengths that are not multiples of eight either way)."""
    eng = make(E, 128)
    rng = np.random.Generator(np.random.PCG64(11))
    total, C = 70_001, 6
    ks = [3_001, 17, 2_048, 0, 999, 4_097]
    locs = [np.sort(rng.choice(total, kc, replace=False)).astype(np.uint32) for kc in ks]
    vals = [rng.integers(0, 2 ** 63, (kc, 2), dtype=np.uint64) for kc in ks]
    dv = [eng.upload(v) if v.size else eng.alloc(16) for v in vals]
    outs = []
    for shift in (0, 4, 8, 12):
        bufs = [eng.alloc(4 * kc + 32) for kc in ks]
        for bf, l in zip(bufs, locs):
            if l.size:
                bf.upload_at(shift, l)
        dl = [bf.ptr + shift for bf in bufs]
        agg, dec = eng.alloc_vec(total), eng.alloc_vec(total)
        bnd = eng.span_bounds(total, dl, ks)
        eng.sparse_aggregate_dev(total, dl, ks, dv, [5] * C, agg, bounds=bnd)
        eng.sparse_decrypt_dev(2, dl, ks, total, 16, agg, dec, bounds=bnd)
        outs.append((agg.download(np.uint64, 2 * total), dec.download(np.uint64, 2 * total)))
        del bnd
    want = oracle.sparse_minus_mask(KEY, 2, locs, total, 16, 128)
    assert np.array_equal(outs[0][1].reshape(total, 2), oracle.combine(128, outs[0][0].reshape(total, 2), None, want))
    for a, d in outs[1:]:
        assert np.array_equal(a, outs[0][0]) and np.array_equal(d, outs[0][1])


@pytest.mark.parametrize("b,C,total,k,parts", [(128, 7, 100_003, 2_000, 3), (128, 70, 50_000, 300, 8), (100, 5, 1_752 * 4, 500, 4), (128, 3, 9_000, 9_000, 5),
                                              (128, 4, 1_000, 100, 3)])
def test_sparse_round_sharded_by_position_ranges(E, oracle, b, C, total, k, parts):
    """flashe_sparse_encrypt_aggregate_range_dev / flashe_sparse_decrypt_range_dev: the dense vector cut into `parts` position ranges
    (span-aligned, the way the GPUs of a node would share a sparse round) -- every range run on its own, its aggregate and decrypt slices
    laid side by side equal the whole-vector calls and the oracle; the ciphertext entries a range writes are exactly the entries whose
    position it owns; ranges that are not span-aligned are refused; empty ranges (more parts than spans) do nothing."""
    from flashe_amd._lib import FlasheError
    eng = make(E, b)
    span = eng.sparse_span()
    rng = np.random.Generator(np.random.PCG64(b + C + parts))
    ks = [k] * C
    if C >= 3:
        ks[1], ks[2] = max(k // 4, 1), 0
    locs = [np.sort(rng.choice(total, kc, replace=False)).astype(np.uint32) for kc in ks]
    pts = [rng.integers(0, 2 ** 62, kc, dtype=np.uint64) for kc in ks]
    idx = [(5 * c + 2) % 89 for c in range(C)]
    zeros = [3 + c for c in range(C)]
    dl = [eng.upload(l) if l.size else eng.alloc(16) for l in locs]
    dp = [eng.upload(p) if p.size else eng.alloc(16) for p in pts]
    bnd = eng.span_bounds(total, dl, ks)
    # the whole vector at once
    cts0 = [eng.alloc_vec(max(kc, 1)) for kc in ks]
    agg0, dec0 = eng.alloc_vec(total), eng.alloc_vec(total)
    eng.sparse_encrypt_aggregate_dev(4, idx, dl, ks, dp, 1, zeros, total, 16, cts0, agg0, bounds=bnd)
    # (the decrypt twin subtracts the masks of prefixes 0 .. C-1: a consistency check of the range form against the whole-vector form)
    eng.sparse_decrypt_dev(4, dl, ks, total, 16, agg0, dec0, sorted_lists=True, bounds=bnd)
    want_agg, want_dec = agg0.download(np.uint64, 2 * total), dec0.download(np.uint64, 2 * total)
    # position ranges: `parts` contiguous runs of whole spans
    n_spans = (total + span - 1) // span
    edges = [min(total, span * ((n_spans * g) // parts)) for g in range(parts)] + [total]
    cts = [eng.alloc_vec(max(kc, 1)) for kc in ks]
    for c_ in cts:
        eng.memset_dev(c_, 0xee, c_.nbytes)
    got_agg, got_dec = np.zeros(2 * total, dtype=np.uint64), np.zeros(2 * total, dtype=np.uint64)
    for g in range(parts):
        first, count = edges[g], edges[g + 1] - edges[g]
        sl_a, sl_d = eng.alloc_vec(max(count, 1)), eng.alloc_vec(max(count, 1))
        eng.sparse_encrypt_aggregate_dev(4, idx, dl, ks, dp, 1, zeros, total, 16, cts, sl_a, bounds=bnd, position_range=(first, count))
        eng.sparse_decrypt_dev(4, dl, ks, total, 16, sl_a, sl_d, bounds=bnd, position_range=(first, count))
        if count:
            got_agg[2 * first:2 * (first + count)] = sl_a.download(np.uint64, 2 * count)
            got_dec[2 * first:2 * (first + count)] = sl_d.download(np.uint64, 2 * count)
        # so far exactly the entries at positions below edges[g + 1] carry their ciphertext
        for c in range(C):
            if ks[c]:
                have = cts[c].download(np.uint64, 2 * ks[c]).reshape(ks[c], 2)
                ref = cts0[c].download(np.uint64, 2 * ks[c]).reshape(ks[c], 2)
                done = locs[c] < edges[g + 1]
                assert np.array_equal(have[done], ref[done]) and np.all(have[~done] == np.uint64(0xeeeeeeeeeeeeeeee)), (g, c)
    assert np.array_equal(got_agg, want_agg) and np.array_equal(got_dec, want_dec), (b, C, parts)
    mask = oracle.sparse_minus_mask(KEY, 4, locs, total, 16, b)
    assert np.array_equal(got_dec.reshape(total, 2), oracle.combine(b, want_agg.reshape(total, 2), None, mask))
    if total > span + 5:
        with pytest.raises(FlasheError):
            eng.sparse_decrypt_dev(4, dl, ks, total, 16, agg0, dec0, bounds=bnd, position_range=(5, span))
        with pytest.raises(FlasheError):
            eng.sparse_decrypt_dev(4, dl, ks, total, 16, agg0, dec0, bounds=bnd, position_range=(0, span + 1))
