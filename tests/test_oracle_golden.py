"""CPU-only: pin the C oracle (oracle/flashe_oracle.c) against the golden vectors that
tests/golden/gen_golden.py produced from the unmodified reference, plus FIPS-197 KATs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, unhex

KEY = bytes(range(32))


def L(b):
    return 2 if b > 64 else 1


def test_fips197_c3_and_key_expansion(oracle):
    # FIPS-197 Appendix C.3
    ct = oracle.aes256_encrypt_block(bytes(range(32)), bytes.fromhex("00112233445566778899aabbccddeeff"))
    assert ct.hex() == "8ea2b7ca516745bfeafc49904b496089"
    # FIPS-197 Appendix A.3 key expansion, first and last words
    key = bytes.fromhex("603deb1015ca71be2b73aef0857d77811f352c073b6108d72d9810a30914dff4")
    rk = oracle.aes256_round_keys(key)
    assert rk[8] == 0x9ba35411 and rk[9] == 0x8e6925af and rk[59] == 0x706c631e
    # SP 800-38A F.1.5 ECB-AES256 block 1
    ct = oracle.aes256_encrypt_block(key, bytes.fromhex("6bc1bee22e409f96e93d7e117393172a"))
    assert ct.hex() == "f3eed1bdb5d2a03c064b5a7e3db181f8"


def test_aes_anchor_blocks(oracle):
    g = load_golden("aes_anchors.json")
    for c in g["blocks"]:
        assert oracle.aes256_encrypt_block(KEY, bytes.fromhex(c["block"])).hex() == c["out"]
    # the survey's probe anchors
    blk = (3).to_bytes(4, "big") + (0).to_bytes(4, "big") + (0).to_bytes(8, "big")
    assert oracle.aes256_encrypt_block(KEY, blk).hex() == "22a519ae4e443095820596936975b11d"


def test_chunks(oracle):
    g = load_golden("mask_streams.json")
    for c in g["cases"]:
        begins = oracle.chunks(c["n"], c["n_jobs"])
        assert [[begins[i], begins[i + 1]] for i in range(c["n_jobs"])] == c["chunks"]


def test_mask_streams(oracle):
    g = load_golden("mask_streams.json")
    for c in g["cases"]:
        got = oracle.limbs_to_ints(oracle.mask(KEY, c["iter"], c["idx"], c["n"], c["n_jobs"], c["b"]))
        assert got == unhex(c["stream"]), (c["b"], c["n"], c["n_jobs"])


def test_mask_sums(oracle):
    g = load_golden("mask_streams.json")
    for c in g["sums"]:
        a = oracle.limbs_to_ints(oracle.mask_sum(KEY, c["iter"], c["add_idx"], c["n"], c["n_jobs"], c["b"]))
        m = oracle.limbs_to_ints(oracle.mask_sum(KEY, c["iter"], c["minus_idx"], c["n"], c["n_jobs"], c["b"]))
        assert a == unhex(c["add"]) and m == unhex(c["minus"])


def test_cipher_rounds(oracle):
    g = load_golden("cipher_rounds.json")
    for c in g["cases"]:
        b, n, J, it, scheme = c["b"], c["n"], c["n_jobs"], c["iter"], c["scheme"]
        cts = {}
        for i, pt in c["pt"].items():
            ptl = oracle.ints_to_limbs(unhex(pt), b).reshape(n, L(b))
            ct = oracle.encrypt(KEY, it, int(i), scheme, J, b, ptl)
            assert oracle.limbs_to_ints(ct) == unhex(c["ct"][i]), (scheme, b, n, i)
            cts[int(i)] = ct
        models = [cts[i] for i in c["uploaded"]]
        agg = oracle.aggregate_elem(models, b)
        assert oracle.limbs_to_ints(agg) == unhex(c["agg_elem"])
        # packed reduce
        packed = [oracle.pack(m, b) for m in models]
        aggp = oracle.aggregate_packed(packed, n * b)
        got_int = sum(int(v) << (64 * i) for i, v in enumerate(aggp))
        assert got_int == int(c["agg_packed_int"], 16)
        aggp_el = oracle.unpack(aggp, n, b)
        assert oracle.limbs_to_ints(aggp_el) == unhex(c["agg_packed"])
        # telescoping + decrypt
        if scheme == "double":
            add_idx, minus_idx = oracle.telescope(c["uploaded"])
            pa = [(it.to_bytes(4, "big") + i.to_bytes(4, "big")).hex() for i in add_idx]
            assert pa == c["prefix_add"]
        else:
            add_idx, minus_idx = [], list(c["uploaded"])
        pm = [(it.to_bytes(4, "big") + i.to_bytes(4, "big")).hex() for i in minus_idx]
        assert pm == c["prefix_minus"]
        dec = oracle.decrypt(KEY, it, add_idx, minus_idx, J, b, agg)
        assert oracle.limbs_to_ints(dec) == unhex(c["dec_elem"])
        decp = oracle.decrypt(KEY, it, add_idx, minus_idx, J, b, aggp_el)
        assert oracle.limbs_to_ints(decp) == unhex(c["dec_packed"])


def test_precompute(oracle):
    g = load_golden("precompute.json")
    for c in g["cases"]:
        b, n, J, it, C = c["b"], c["n"], c["n_jobs"], c["iter"], c["num_clients"]
        for i, cl in c["clients"].items():
            add = oracle.mask(KEY, it, int(i), n, J, b)
            minus = oracle.mask(KEY, it, int(i) + 1, n, J, b)
            assert oracle.limbs_to_ints(add) == unhex(cl["pre_add"])
            assert oracle.limbs_to_ints(minus) == unhex(cl["pre_minus"])
            pt = oracle.ints_to_limbs(unhex(c["pt"][i]), b)
            assert oracle.limbs_to_ints(oracle.combine(b, pt, add, minus)) == unhex(cl["ct"])
        padd = oracle.mask(KEY, it, C, n, J, b)
        pminus = oracle.mask(KEY, it, 0, n, J, b)
        assert oracle.limbs_to_ints(padd) == unhex(c["dec_pre_add"])
        assert oracle.limbs_to_ints(pminus) == unhex(c["dec_pre_minus"])
        ea = [int(p[8:], 16) for p in c["extra_prefix_add"]]
        em = [int(p[8:], 16) for p in c["extra_prefix_minus"]]
        xa = oracle.combine(b, padd, oracle.mask_sum(KEY, it, ea, n, J, b), None)
        xm = oracle.combine(b, pminus, oracle.mask_sum(KEY, it, em, n, J, b), None)
        agg = oracle.ints_to_limbs(unhex(c["agg"]), b)
        assert oracle.limbs_to_ints(oracle.combine(b, agg, xa, xm)) == unhex(c["dec"])


def test_pack_unpack(oracle):
    g = load_golden("pack.json")
    for c in g["cases"] + g["merges"]:
        b, n = c["b"], c["n"]
        v = oracle.ints_to_limbs(unhex(c["vals"]), b)
        p = oracle.pack(v, b)
        assert sum(int(x) << (64 * i) for i, x in enumerate(p)) == int(c["packed_int"], 16)
        assert oracle.limbs_to_ints(oracle.unpack(p, n, b)) == unhex(c["vals"])
    ce = g["carry_example"]
    a = oracle.ints_to_limbs(unhex(ce["a"]), 8)
    c2 = oracle.ints_to_limbs(unhex(ce["c"]), 8)
    s = oracle.aggregate_packed([oracle.pack(a, 8), oracle.pack(c2, 8)], 24)
    assert oracle.limbs_to_ints(oracle.unpack(s, 3, 8)) == unhex(ce["packed_sum"])
    assert oracle.limbs_to_ints(oracle.aggregate_elem([a, c2], 8)) == unhex(ce["elem_sum"])


def test_sparse_single(oracle):
    g = load_golden("sparse.json")
    for c in g["single"]:
        b, total, J, it, C = c["b"], c["total"], c["n_jobs"], c["iter"], c["num_clients"]
        dense = []
        for i in range(C):
            pt = oracle.ints_to_limbs(unhex(c["pt"][i]), b)
            ct = oracle.encrypt(KEY, it, i, "single", J, b, pt)
            up = unhex(c["uploads"][i])
            assert oracle.limbs_to_ints(ct) == up[:-1]
            zero = oracle.ints_to_limbs([up[-1]], b)
            d = oracle.expand_to_dense(total, c["locs"][i], ct, zero, b)
            assert oracle.limbs_to_ints(d) == unhex(c["dense"][i])
            dense.append(d)
        agg = oracle.aggregate_elem(dense, b)
        assert oracle.limbs_to_ints(agg) == unhex(c["agg"])
        mm = oracle.sparse_minus_mask(KEY, it, c["locs"], total, J, b)
        assert oracle.limbs_to_ints(mm) == unhex(c["minus_mask"])
        assert oracle.limbs_to_ints(oracle.combine(b, agg, None, mm)) == unhex(c["dec"])


def test_sparse_dense_double(oracle):
    g = load_golden("sparse.json")
    for c in g["dense_double"]:
        a = oracle.sparse_dense_mask(KEY, c["iter"], c["add_sel"], c["total"], c["b"])
        m = oracle.sparse_dense_mask(KEY, c["iter"], c["minus_sel"], c["total"], c["b"])
        assert oracle.limbs_to_ints(a) == unhex(c["add"])
        assert oracle.limbs_to_ints(m) == unhex(c["minus"])


def test_config1_plumbing(oracle):
    """BASELINE config 1: 1e4 fp32 -> 32-bit quantise -> 64-bit modulus, 2 clients, single mask."""
    z = np.load(os.path.join(GOLDEN, "config1.npz"))
    n, b, J = 10000, 64, 8
    cts = []
    for c in range(2):
        ct = oracle.encrypt(KEY, 0, c, "single", J, b, z[f"q{c}"])
        assert np.array_equal(ct[:, 0], z[f"ct{c}"])
        cts.append(ct)
    agg = oracle.aggregate_elem(cts, b)
    assert np.array_equal(agg[:, 0], z["agg_elem"])
    aggp = oracle.unpack(oracle.aggregate_packed([oracle.pack(c, b) for c in cts], n * b), n, b)
    assert np.array_equal(aggp[:, 0], z["agg_packed"])
    assert int((aggp[:, 0] != agg[:, 0]).sum()) == 4969          # SURVEY.md 8(d) probe
    dec = oracle.decrypt(KEY, 0, [], [0, 1], J, b, agg)
    assert np.array_equal(dec[:, 0], z["dec_elem"])
    assert np.array_equal(dec[:, 0], z["q0"] + z["q1"])
    decp = oracle.decrypt(KEY, 0, [], [0, 1], J, b, aggp)
    assert np.array_equal(decp[:, 0], z["dec_packed"])
    # unquantise (jzf_quantize.py:102-107) restated: both aggregates land on the clipped float sum
    alpha = float(z["alpha"]) * 2
    for d, ref in ((dec, z["unq_elem"]), (decp, z["unq_packed"])):
        unq = d[:, 0].astype(np.float64) * (2 * alpha) / (((1 << 32) - 1) * 2) - alpha
        assert np.allclose(unq, ref, rtol=0, atol=1e-9)
    true = np.clip(z["x0"], -alpha / 2, alpha / 2).astype(np.float64) + np.clip(z["x1"], -alpha / 2, alpha / 2)
    assert np.abs(z["unq_elem"] - true).max() < 1e-5 and np.abs(z["unq_packed"] - true).max() < 1e-5


def test_quantise_codec(oracle):
    """quantise / batch / unbatch / unquantise against the reference's outputs."""
    g = load_golden("codec.json")
    for c in g["quantize"]:
        x = np.frombuffer(bytes.fromhex(c["x"]), dtype=c["dtype"])
        u = np.frombuffer(bytes.fromhex(c["u"]), dtype=np.float64)
        q = oracle.quantize(x, float.fromhex(c["alpha"]), c["element_bits"], u)
        assert [int(v) for v in q] == unhex(c["q"]), (c["dtype"], c["element_bits"])
    for c in g["batch"]:
        fb = c["element_bits"] + c["factor"]
        vals = np.array(unhex(c["vals"]), dtype=np.uint64)
        b = oracle.batch(vals, c["int_bits"], fb)
        assert oracle.limbs_to_ints(b) == unhex(c["batched"])
        assert [int(v) for v in oracle.unbatch(b, c["int_bits"], fb)] == unhex(c["unbatched"])
    for c in g["unquantize"]:
        vals = oracle.ints_to_limbs(unhex(c["vals"]), 128)
        want = np.frombuffer(bytes.fromhex(c["out"]), dtype=np.float64)
        got = oracle.unquantize(vals, float.fromhex(c["alpha"]), c["element_bits"], c["num_clients"])
        assert got.tobytes() == want.tobytes(), c["element_bits"]


def test_sparsify(oracle):
    """per-layer top-k with residual accumulation, two consecutive rounds."""
    for c in load_golden("sparsify.json")["cases"]:
        dt = np.dtype(c["dtype"])
        remain = np.zeros(c["n"], dtype=dt)
        for rd in c["rounds"]:
            layer = np.frombuffer(bytes.fromhex(rd["layer"]), dtype=dt)
            loc, vals, remain = oracle.sparsify(layer, rd["k"], remain)
            assert [int(v) for v in loc] == rd["location"]
            assert vals.tobytes().hex() == rd["masked"] and remain.tobytes().hex() == rd["remain"]
    # ties at the threshold go to the higher index (stable argsort semantics)
    layer = np.array([1, -2, 2, 0.5, 2, -1], dtype=np.float32)
    loc, vals, rem = oracle.sparsify(layer, 2, np.zeros(6, dtype=np.float32))
    want = sorted(np.abs(layer).argsort(kind="stable")[-2:].tolist())
    assert [int(v) for v in loc] == want == [2, 4]


def test_aesni_and_table_paths_agree(oracle):
    """The AES-NI fast path of the oracle (used for the CPU baseline) against its portable table path."""
    if not oracle.aesni_available():
        pytest.skip("no AES-NI on this CPU / compiler")
    try:
        for b, n, J in [(128, 4099, 3), (64, 5000, 8), (20, 7001, 16)]:
            pt = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            if b < 64:
                pt &= np.uint64((1 << b) - 1)
            oracle.set_aesni(False)
            a = oracle.encrypt(KEY, 5, 7, "double", J, b, pt)
            m = oracle.mask(KEY, 5, 9, n, J, b)
            oracle.set_aesni(True)
            for vaes in (False, True):                 # 8-way AES-NI, then (when the CPU has it) 16-way VAES
                oracle.set_vaes(vaes)
                assert np.array_equal(a, oracle.encrypt(KEY, 5, 7, "double", J, b, pt)), (b, vaes)
                assert np.array_equal(m, oracle.mask(KEY, 5, 9, n, J, b)), (b, vaes)
    finally:
        oracle.set_aesni(True)
        oracle.set_vaes(True)


def test_telescope_examples(oracle):
    assert oracle.telescope([0, 1, 2, 4]) == ([3, 5], [0, 4])
    assert oracle.telescope([0] * 4) == ([1, 1, 1, 1], [0, 0, 0, 0])
    assert oracle.telescope([5, 0, 2, 3]) == ([1, 4, 6], [0, 2, 5])
    assert oracle.telescope([]) == ([], [])
