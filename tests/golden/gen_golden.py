#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference); the fixtures it writes are
plain data (inputs + expected outputs) and are what travels to the GPU box.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.json / *.npz

How the reference is imported: the reference source is used as
is; only its missing third-party imports are shimmed in sys.modules --
  * cachetools / compress_pickle: not on the path we call, stubbed so imports succeed;
  * Crypto.Cipher.AES (pycryptodome 3.9.9, absent here): AES.new(key, MODE_ECB)
    .encrypt(block) is forwarded to the system libcrypto 3 EVP aes-256-ecb (no padding)
    through ctypes.  AES-256 is FIPS-197, so any conformant implementation gives the
    same bytes; the shim itself is checked against FIPS-197 C.3 below.
jzf_aggregator.py imports sibling modules that are not in the tree (jzf_additive_mask_block,
jzf_simple_block) or need absent libraries (bfv / ckks / paillier blocks): those get empty
stand-ins in sys.modules (_import_aggregator), after which the module imports and its
Client.sparsify (:578-623) and Arbiter.expand_to_dense (:150-165) are CALLED, unbound, on stub
objects.  The two arbiter reduces (:406-419, :424-430) are inline expressions of a long method
and are evaluated here exactly as written there: reduce(lambda x, y: (x + y) % mod, models) on
Python ints / object arrays.
"""
import collections
import collections.abc
import ctypes
import ctypes.util
import json
import os
import sys
import types
from functools import reduce

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
KEY = bytes(range(32))


# --------------------------------------------------------------------------- shims
def install_shims():
    sys.dont_write_bytecode = True
    ct = types.ModuleType("cachetools")

    class LRUCache(dict):
        def __init__(self, maxsize=None, *a, **k):
            super().__init__()

    def cached(cache=None, **_):
        return lambda f: f

    ct.LRUCache, ct.cached = LRUCache, cached
    sys.modules["cachetools"] = ct
    sys.modules["compress_pickle"] = types.ModuleType("compress_pickle")
    collections.Iterable = collections.abc.Iterable

    crypto = ctypes.CDLL(ctypes.util.find_library("crypto"))
    crypto.EVP_CIPHER_CTX_new.restype = ctypes.c_void_p
    crypto.EVP_aes_256_ecb.restype = ctypes.c_void_p
    crypto.EVP_EncryptInit_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_char_p, ctypes.c_char_p]
    crypto.EVP_CIPHER_CTX_set_padding.argtypes = [ctypes.c_void_p, ctypes.c_int]
    crypto.EVP_EncryptUpdate.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int),
                                         ctypes.c_char_p, ctypes.c_int]

    class _Ecb:
        def __init__(self, key):
            assert len(key) == 32
            self.ctx = crypto.EVP_CIPHER_CTX_new()
            assert crypto.EVP_EncryptInit_ex(self.ctx, crypto.EVP_aes_256_ecb(), None, key, None) == 1
            crypto.EVP_CIPHER_CTX_set_padding(self.ctx, 0)

        def encrypt(self, data):
            out = ctypes.create_string_buffer(len(data) + 16)
            n = ctypes.c_int(0)
            assert crypto.EVP_EncryptUpdate(self.ctx, out, ctypes.byref(n), bytes(data), len(data)) == 1
            return out.raw[:n.value]

    aes = types.ModuleType("Crypto.Cipher.AES")
    aes.MODE_ECB, aes.MODE_CTR = 1, 6

    def new(key, mode, **kw):
        if mode != aes.MODE_ECB:
            raise NotImplementedError("only ECB is on the hot path")
        return _Ecb(key)

    aes.new = new
    m_crypto = types.ModuleType("Crypto")
    m_cipher = types.ModuleType("Crypto.Cipher")
    m_util = types.ModuleType("Crypto.Util")
    m_counter = types.ModuleType("Crypto.Util.Counter")
    m_counter.new = lambda *a, **k: None
    m_cipher.AES = aes
    m_util.Counter = m_counter
    m_crypto.Cipher, m_crypto.Util = m_cipher, m_util
    sys.modules.update({"Crypto": m_crypto, "Crypto.Cipher": m_cipher, "Crypto.Cipher.AES": aes,
                        "Crypto.Util": m_util, "Crypto.Util.Counter": m_counter})
    sys.path.insert(0, REF)
    # FIPS-197 Appendix C.3 through the shim
    got = _Ecb(bytes(range(32))).encrypt(bytes.fromhex("00112233445566778899aabbccddeeff")).hex()
    assert got == "8ea2b7ca516745bfeafc49904b496089", got


install_shims()
from federatedml.secureprotol import jzf_flashe as RF                      # noqa: E402
from federatedml.secureprotol.jzf_aes_prp import PsuedoRandomPermutation   # noqa: E402
from federatedml.framework import jzf_weights as RW                        # noqa: E402
from federatedml.secureprotol import jzf_quantize as RQ                    # noqa: E402
from federatedml.secureprotol.jzf_aciq import ACIQ                         # noqa: E402


def hx(v):
    return format(int(v), "x")


def hxl(vs):
    return [hx(v) for v in vs]


def dump(name, obj):
    p = os.path.join(HERE, name)
    with open(p, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {name}: {os.path.getsize(p)} bytes")


def new_cipher(b, scheme, idx, it, num_clients=None):
    c = RF.FlasheCipher(b, mask=scheme)
    if num_clients is not None:
        c.set_num_clients(num_clients)
    c.generate_prp_seed(KEY)
    c.set_iter_index(it)
    c.idx = idx
    return c


def rand_ints(rng, n, bits):
    return [int.from_bytes(rng.bytes(16), "little") & ((1 << bits) - 1) for _ in range(n)]


# --------------------------------------------------------------------------- AES anchors
def gen_aes():
    prp = PsuedoRandomPermutation()
    prp.generate_key(assigned_key=KEY)
    cases = []
    for it, idx, ctr in [(3, 0, 0), (3, 0, 6), (3, 0, 13), (3, 1, 0), (0, 0, 0), (0, 10, 9999999),
                         (2 ** 32 - 1, 2 ** 32 - 1, 2 ** 64 - 1), (7, 5, 2 ** 32), (1, 2, 2 ** 40 + 12345)]:
        block = it.to_bytes(4, "big") + idx.to_bytes(4, "big") + ctr.to_bytes(8, "big")
        cases.append({"iter": it, "idx": idx, "counter": str(ctr), "block": block.hex(),
                      "out": prp.get_permutation(block).hex()})
    # key normalisation (jzf_aes.py:21-28): int and over-long bytes seeds collapse to the low 32 bytes
    norm = []
    for seed in [KEY, b"\x00" * 224 + KEY, b"\xaa" * 7 + KEY]:
        c = RF.FlasheCipher(128)
        c.generate_prp_seed(seed)
        norm.append({"seed": seed.hex(), "aes_key": c.prp.aes.get_key().hex(),
                     "prp_seed_len": len(c.get_prp_seed())})
    c = RF.FlasheCipher(128)
    c.generate_prp_seed(int.from_bytes(KEY, "big"))
    norm.append({"seed_int": hx(int.from_bytes(KEY, "big")), "aes_key": c.prp.aes.get_key().hex(),
                 "prp_seed_len": len(c.get_prp_seed())})
    dump("aes_anchors.json", {"key": KEY.hex(), "fips197_c3": {
        "key": bytes(range(32)).hex(), "pt": "00112233445566778899aabbccddeeff",
        "ct": "8ea2b7ca516745bfeafc49904b496089"}, "blocks": cases, "key_norm": norm})


# --------------------------------------------------------------------------- mask streams
def ref_stream(b, it, idx, n, n_jobs):
    """What _multiprocessing_encrypt_single concatenates (jzf_flashe.py:436-445), in-process."""
    prefix = it.to_bytes(4, "big") + idx.to_bytes(4, "big")
    out = []
    for begin, end in RF.chunks_idx(range(n), n_jobs):
        out += RF._static_prepare_encrypt_single(begin, end, KEY, b, prefix)
    return out


def gen_masks():
    cases = []
    for b in [1, 7, 8, 20, 23, 32, 33, 63, 64, 65, 100, 120, 127, 128]:
        for n, n_jobs in [(1, 1), (5, 3), (7, 1), (50, 1), (50, 4), (50, 16), (131, 8), (3, 16)]:
            it, idx = 3, (0 if b % 2 else 1)
            cases.append({"b": b, "n": n, "n_jobs": n_jobs, "iter": it, "idx": idx,
                          "chunks": [list(x) for x in RF.chunks_idx(range(n), n_jobs)],
                          "stream": hxl(ref_stream(b, it, idx, n, n_jobs))})
    # one larger case with ragged chunks
    for b, n, n_jobs in [(20, 4099, 16), (64, 4099, 8), (128, 4099, 3), (23, 1000, 16)]:
        cases.append({"b": b, "n": n, "n_jobs": n_jobs, "iter": 11, "idx": 9,
                      "chunks": [list(x) for x in RF.chunks_idx(range(n), n_jobs)],
                      "stream": hxl(ref_stream(b, 11, 9, n, n_jobs))})
    # multi-prefix sums, both halves of _static_prepare_decrypt (jzf_flashe.py:115-152)
    sums = []
    for b, n, n_jobs in [(128, 9, 2), (64, 21, 4), (20, 40, 3), (7, 100, 8)]:
        add_idx, minus_idx = [3, 5, 5], [0, 4]
        pa = [(2).to_bytes(4, "big") + i.to_bytes(4, "big") for i in add_idx]
        pm = [(2).to_bytes(4, "big") + i.to_bytes(4, "big") for i in minus_idx]
        A, M = [], []
        for begin, end in RF.chunks_idx(range(n), n_jobs):
            a, m = RF._static_prepare_decrypt(begin, end, KEY, b, pa, pm)
            A += a
            M += m
        S = []
        for begin, end in RF.chunks_idx(range(n), n_jobs):
            S += RF._static_prepare_decrypt_single(begin, end, KEY, b, pm)
        assert S == M
        sums.append({"b": b, "n": n, "n_jobs": n_jobs, "iter": 2, "add_idx": add_idx,
                     "minus_idx": minus_idx, "add": hxl(A), "minus": hxl(M)})
    dump("mask_streams.json", {"key": KEY.hex(), "cases": cases, "sums": sums})


# --------------------------------------------------------------------------- full cipher rounds
def packed_int(vals, b):
    return RW._to_bytes_old(list(vals), b)[0]


def gen_rounds():
    rng = np.random.RandomState(20240923)
    cases = []
    grid = [
        # (scheme, b, pt_bits, n, n_jobs, C, uploaded idx list, iter)
        ("double", 128, 64, 37, 1, 3, [0, 1, 2], 0),
        ("double", 128, 64, 37, 8, 5, [0, 1, 2, 4], 3),          # dropout of client 3
        ("double", 128, 128, 16, 3, 4, [0, 0, 0, 0], 1),         # notebook-style duplicates
        ("double", 120, 100, 50, 16, 6, [5, 0, 2, 3], 7),        # unsorted, two gaps
        ("double", 64, 60, 50, 4, 3, [0, 1, 2], 2),
        ("double", 64, 64, 131, 8, 2, [1], 2),
        ("double", 20, 16, 257, 16, 10, list(range(10)), 0),
        ("double", 23, 16, 100, 16, 4, [0, 1, 3], 5),
        ("double", 7, 4, 64, 3, 3, [0, 2], 9),
        ("double", 1, 1, 40, 2, 2, [0, 1], 0),
        ("double", 33, 30, 19, 5, 2, [0, 1], 4),
        ("double", 65, 64, 19, 5, 2, [0, 1], 4),
        ("single", 64, 34, 100, 8, 2, [0, 1], 0),
        ("single", 128, 64, 21, 2, 4, [0, 2, 3], 6),
        ("single", 20, 16, 90, 16, 3, [2, 0, 1], 1),
        ("double", 128, 64, 0, 4, 2, [0, 1], 0),                  # empty vector
        ("double", 128, 64, 3, 16, 2, [0, 1], 0),                 # n < n_jobs: empty chunks
    ]
    for scheme, b, pt_bits, n, n_jobs, C, up, it in grid:
        RF.N_JOBS = n_jobs
        pts, cts = {}, {}
        for i in sorted(set(up)):
            pts[i] = rand_ints(rng, n, pt_bits)
            c = new_cipher(b, scheme, i, it, C)
            out = c.encrypt(np.array(pts[i], dtype=object))
            cts[i] = [int(x) for x in out]
        models = [np.array(cts[i], dtype=object) for i in up]
        mod = 1 << b
        agg_elem = reduce(lambda x, y: (x + y) % mod, models) if n else np.array([], dtype=object)
        if len(models) == 1:
            agg_elem = models[0] % mod
        # packed reduce, jzf_aggregator.py:406-419
        if n:
            pmod = 1 << (b * n)
            pk = [packed_int(cts[i], b) for i in up]
            agg_packed_int = reduce(lambda x, y: (x + y) % pmod, pk) if len(pk) > 1 else pk[0] % pmod
            agg_packed = RW._from_bytes_old(agg_packed_int, n, b)
            agg_packed.reverse()
        else:
            agg_packed_int, agg_packed = 0, []
        d = new_cipher(b, scheme, 0, it, C)
        raw = list(up)
        d.set_idx_list(raw_idx_list=raw, mode="decrypt")
        pre_add = [p.hex() for p in d.index_prefix_for_add] if scheme == "double" else []
        pre_minus = [p.hex() for p in d.index_prefix_for_minus]
        dec_elem = d.decrypt(np.array([int(x) for x in agg_elem], dtype=object)) if n else []
        d2 = new_cipher(b, scheme, 0, it, C)
        d2.set_idx_list(raw_idx_list=list(up), mode="decrypt")
        dec_packed = d2.decrypt(np.array(agg_packed, dtype=object)) if n else []
        true_sum = [sum(pts[i][j] for i in up) % mod for j in range(n)]
        if n:
            assert [int(x) for x in dec_elem] == true_sum, (scheme, b, n)
        cases.append({
            "scheme": scheme, "b": b, "n": n, "n_jobs": n_jobs, "num_clients": C, "uploaded": up,
            "iter": it, "pt": {str(i): hxl(v) for i, v in pts.items()},
            "ct": {str(i): hxl(v) for i, v in cts.items()},
            "agg_elem": hxl(agg_elem), "agg_packed_int": hx(agg_packed_int),
            "agg_packed": hxl(agg_packed), "prefix_add": pre_add, "prefix_minus": pre_minus,
            "dec_elem": hxl(dec_elem), "dec_packed": hxl(dec_packed)})
    dump("cipher_rounds.json", {"key": KEY.hex(), "cases": cases})


# --------------------------------------------------------------------------- precompute
def gen_precompute():
    rng = np.random.RandomState(77)
    cases = []
    for b, n, n_jobs, C, up in [(128, 29, 4, 3, [0, 1, 2]), (23, 61, 16, 4, [0, 1, 3]), (64, 33, 2, 5, [1, 2, 3])]:
        RF.N_JOBS = n_jobs
        it = 4
        pts, cts = {}, {}
        for i in sorted(set(up)):
            pts[i] = rand_ints(rng, n, min(b, 64) - 3)
            c = new_cipher(b, "double", i, it - 1, C)
            c.set_num_params(n)
            c.prepare_encrypt()                       # masks for iteration `it`
            pre_add = hxl(c.next_iter_encrypt_prepared["add"])
            pre_minus = hxl(c.next_iter_encrypt_prepared["minus"])
            c.set_iter_index(it)
            out = c.encrypt(np.array(pts[i], dtype=object))
            assert c.next_iter_encrypt_prepared == {}
            cts[i] = {"ct": hxl(out), "pre_add": pre_add, "pre_minus": pre_minus}
        mod = 1 << b
        agg = reduce(lambda x, y: (x + y) % mod,
                     [np.array([int(v, 16) for v in cts[i]["ct"]], dtype=object) for i in up])
        d = new_cipher(b, "double", 0, it, C)
        d.set_num_params(n)
        d.prepare_decrypt()
        dpre_add = hxl(d.next_iter_decrypt_prepared["add"])
        dpre_minus = hxl(d.next_iter_decrypt_prepared["minus"])
        d.set_idx_list(raw_idx_list=list(up), mode="decrypt")
        extra_add = [p.hex() for p in d.index_prefix_for_add]
        extra_minus = [p.hex() for p in d.index_prefix_for_minus]
        dec = d.decrypt(agg)
        # NB the reference applies the precomputed {add: C, minus: 0} masks unconditionally
        # (jzf_flashe.py:557-571), so when client 0 or client C-1 dropped out the result is NOT
        # the plaintext sum.  The fixture records what the reference returns either way.
        ok = [int(x) for x in dec] == [sum(pts[i][j] for i in up) % mod for j in range(n)]
        assert ok == (0 in up and C - 1 in up), (b, up)
        cases.append({"roundtrip": ok, "b": b, "n": n, "n_jobs": n_jobs, "num_clients": C, "uploaded": up, "iter": it,
                      "pt": {str(i): hxl(v) for i, v in pts.items()}, "clients": {str(i): v for i, v in cts.items()},
                      "agg": hxl(agg), "dec_pre_add": dpre_add, "dec_pre_minus": dpre_minus,
                      "extra_prefix_add": extra_add, "extra_prefix_minus": extra_minus, "dec": hxl(dec)})
    dump("precompute.json", {"key": KEY.hex(), "cases": cases})


# --------------------------------------------------------------------------- bit packing
def gen_pack():
    rng = np.random.RandomState(5)
    cases = []
    for b, n in [(1, 70), (7, 33), (8, 9), (20, 13), (23, 50), (32, 5), (33, 7), (64, 6), (65, 5), (120, 9), (128, 4),
                 (20, 1), (128, 1)]:
        v = rand_ints(rng, n, b)
        big, l = RW._to_bytes_old(v, b)
        assert l == n
        back = RW._from_bytes_old(big, n, b)
        back.reverse()
        assert back == v
        cases.append({"b": b, "n": n, "vals": hxl(v), "packed_int": hx(big)})
    # compress()'s chunk merge (jzf_weights.py:171-182) restated on _to_bytes_old outputs
    merges = []
    for b, n, n_jobs in [(20, 50, 4), (23, 61, 16), (128, 10, 3)]:
        v = rand_ints(rng, n, b)
        sizes, outs = [], []
        for begin, end in RW.chunks_idx(range(n), n_jobs):
            sizes.append(end - begin)
            outs.append(RW._to_bytes_old(v[begin:end], b)[0])
        s = 0
        for i, o in enumerate(outs):
            s += o << (int(np.sum(sizes[i + 1:])) * b)
        assert s == RW._to_bytes_old(v, b)[0]
        merges.append({"b": b, "n": n, "n_jobs": n_jobs, "vals": hxl(v), "packed_int": hx(s)})
    # the survey's carry example (b = 8)
    a, c = [0xff, 0x01, 0x80], [0x00, 0xff, 0x90]
    pa, pc = RW._to_bytes_old(a, 8)[0], RW._to_bytes_old(c, 8)[0]
    tot = (pa + pc) % (1 << 24)
    un = RW._from_bytes_old(tot, 3, 8)
    un.reverse()
    carry = {"b": 8, "a": hxl(a), "c": hxl(c), "packed_sum": hxl(un),
             "elem_sum": hxl([(x + y) % 256 for x, y in zip(a, c)])}
    dump("pack.json", {"cases": cases, "merges": merges, "carry_example": carry})


# --------------------------------------------------------------------------- sparse
def gen_sparse():
    RA = _import_aggregator()
    rng = np.random.RandomState(9)
    cases = []
    for b, total, n_jobs, C, frac, it in [(128, 200, 4, 3, 0.2, 2), (64, 301, 8, 4, 0.1, 0), (20, 150, 3, 2, 0.3, 5)]:
        RF.N_JOBS = n_jobs
        mod = 1 << b
        locs, pts, ups, zeros = [], [], [], []
        for c in range(C):
            k = max(1, int(total * frac))
            loc = sorted(rng.choice(total, size=k, replace=False).tolist())
            locs.append(loc)
            pt = rand_ints(rng, k, min(b, 64) - 4)
            pts.append(pt)
            ci = new_cipher(b, "single", c, it, C)
            ct = [int(x) for x in ci.encrypt(np.array(pt, dtype=object))]     # compact positions
            zero = int(rng.randint(0, 1 << 15))                               # trailing plain value
            zeros.append(zero)
            ups.append(ct + [zero])
        # Arbiter.expand_to_dense (jzf_aggregator.py:150-165), the reference's own method, then the element-wise reduce
        dense = ref_expand_to_dense(RA, ups, locs, total)
        agg = reduce(lambda x, y: (x + y) % mod, dense)
        d = new_cipher(b, "single", 0, it, C)
        d.masks = [list(l) for l in locs]
        d.total = total
        d.set_idx_list(raw_idx_list=None, mode="decrypt")
        minus = [int(x) for x in d.next_iter_decrypt_prepared["minus"]]
        dec = [int(x) for x in d.decrypt(agg)]
        exp = [0] * total
        for c in range(C):
            inloc = set(locs[c])
            for p in range(total):
                if p not in inloc:
                    exp[p] = (exp[p] + zeros[c]) % mod
            for q, p in enumerate(locs[c]):
                exp[p] = (exp[p] + pts[c][q]) % mod
        assert dec == exp
        cases.append({"b": b, "total": total, "n_jobs": n_jobs, "num_clients": C, "iter": it, "locs": locs,
                      "pt": [hxl(p) for p in pts], "uploads": [hxl(u) for u in ups],
                      "dense": [hxl(e) for e in dense], "agg": hxl(agg), "minus_mask": hxl(minus),
                      "dec": hxl(dec)})
    # dense-position double-mask sparse masks: direct call of _static_prepare_decrypt_spar with one chunk
    dcases = []
    for b, total, C, it in [(128, 60, 3, 1), (20, 77, 4, 3)]:
        one_hots = []
        for c in range(C):
            a = np.zeros(total, dtype=object)
            a[sorted(rng.choice(total, size=total // 3, replace=False).tolist())] = 1
            one_hots.append(a)
        minus, add = [], [np.zeros(total, dtype=object)]
        for c in range(C):          # jzf_flashe.py:398-407
            minus.append(one_hots[c] & ~one_hots[c - 1] if c > 0 else one_hots[c])
            add.append(one_hots[c] & ~one_hots[c + 1] if c < C - 1 else one_hots[c])
        ta, tm = RF._static_prepare_decrypt_spar(0, total, KEY, b, it.to_bytes(4, "big"), add, minus)
        dcases.append({"b": b, "total": total, "iter": it,
                       "add_sel": [[int(x) & 1 for x in a] for a in add],
                       "minus_sel": [[int(x) & 1 for x in m] for m in minus],
                       "add": hxl(ta), "minus": hxl(tm)})
    dump("sparse.json", {"key": KEY.hex(), "single": cases, "dense_double": dcases})


# --------------------------------------------------------------------------- config 1 (plumbing)
def gen_config1():
    """BASELINE config 1 / SURVEY 8(d): 1e4 fp32, 32-bit quantise, 64-bit modulus, 2 clients,
    single mask, n_jobs = 8.  Arrays go to an .npz (uint64)."""
    n, C, b, eb, n_jobs, it = 10000, 2, 64, 32, 8, 0
    RF.N_JOBS = n_jobs
    alpha = ACIQ(eb).get_alpha_gaus_direct(1.0)
    out = {"alpha": np.float64(alpha)}
    qs, cts = [], []
    for c in range(C):
        x = np.random.RandomState(100 + c).standard_normal(n).astype(np.float32)
        np.random.seed(7 + c)
        q = RQ._static_quantize_padding_asymmetric(x, alpha, eb)
        q = np.array([int(v) for v in q], dtype=object)
        ci = new_cipher(b, "single", c, it, C)
        ct = ci.encrypt(q)
        out[f"x{c}"] = x
        out[f"q{c}"] = np.array([int(v) for v in q], dtype=np.uint64)
        out[f"ct{c}"] = np.array([int(v) for v in ct], dtype=np.uint64)
        qs.append(q)
        cts.append(ct)
    mod = 1 << b
    agg = reduce(lambda x, y: (x + y) % mod, cts)
    pmod = 1 << (b * n)
    pk = [RW._to_bytes_old([int(v) for v in ct], b)[0] for ct in cts]
    aggp = RW._from_bytes_old(reduce(lambda x, y: (x + y) % pmod, pk), n, b)
    aggp.reverse()
    d = new_cipher(b, "single", 0, it, C)
    d.set_idx_list(raw_idx_list=[0, 1], mode="decrypt")
    dec = d.decrypt(agg)
    d2 = new_cipher(b, "single", 0, it, C)
    d2.set_idx_list(raw_idx_list=[0, 1], mode="decrypt")
    decp = d2.decrypt(np.array(aggp, dtype=object))
    out["agg_elem"] = np.array([int(v) for v in agg], dtype=np.uint64)
    out["agg_packed"] = np.array([int(v) for v in aggp], dtype=np.uint64)
    out["dec_elem"] = np.array([int(v) for v in dec], dtype=np.uint64)
    out["dec_packed"] = np.array([int(v) for v in decp], dtype=np.uint64)
    out["unq_elem"] = RQ._static_unquantize_padding_asymmetric(
        np.array([int(v) for v in dec], dtype=object), alpha, eb, C).astype(np.float64)
    out["unq_packed"] = RQ._static_unquantize_padding_asymmetric(
        np.array([int(v) for v in decp], dtype=object), alpha, eb, C).astype(np.float64)
    assert [int(v) for v in dec] == [(int(a) + int(c)) % mod for a, c in zip(qs[0], qs[1])]
    p = os.path.join(HERE, "config1.npz")
    np.savez_compressed(p, **out)
    print(f"wrote config1.npz: {os.path.getsize(p)} bytes; q0[:3] = {[int(v) for v in qs[0][:3]]}")


# --------------------------------------------------------------------------- quantise codec (f-1)
def gen_codec():
    """_static_{quantize,unquantize,batching,unbatching}_padding_asymmetric (jzf_quantize.py:55-67,
    :102-107, :162-185, :234-251) on seeded inputs.  The stochastic-rounding uniforms are numpy's
    legacy global MT19937 stream (np.random.seed(s); np.random.random(n)); they are stored so that
    a port can consume the same draws.  Float arrays are stored as raw little-endian bytes (hex)."""
    out = {"quantize": [], "batch": [], "unquantize": []}
    for dtype, eb, n, seed, scale in [("float32", 16, 257, 11, 1.0), ("float32", 32, 300, 12, 3.0), ("float32", 8, 100, 13, 0.2),
                                      ("float64", 16, 200, 14, 1.0), ("float64", 32, 129, 15, 2.5), ("float32", 20, 64, 16, 10.0)]:
        x = (np.random.RandomState(seed).standard_normal(n) * scale).astype(dtype)
        x[:3] = [0.0, 1e9, -1e9]                                   # exact zero and both clip rails
        alpha = ACIQ(eb).get_alpha_gaus_direct(1.0)
        np.random.seed(1000 + seed)
        u = np.random.random(n)
        np.random.seed(1000 + seed)
        q = RQ._static_quantize_padding_asymmetric(x, alpha, eb)
        out["quantize"].append({"dtype": dtype, "element_bits": eb, "n": n, "alpha": float(alpha).hex(),
                                "x": x.tobytes().hex(), "u": u.tobytes().hex(), "q": hxl(q)})
    rng = np.random.RandomState(99)
    for int_bits, eb, C, n in [(120, 16, 10, 37), (128, 16, 10, 50), (64, 16, 2, 10), (20, 16, 10, 7), (128, 32, 100, 33),
                               (120, 16, 10, 6), (64, 8, 4, 100)]:
        factor = int(np.ceil(np.log2(C)))
        vals = [int(v) for v in rng.randint(0, 2 ** eb, size=n)]
        b = RQ._static_batching_padding_asymmetric(np.array(vals, dtype=object), int_bits, eb, factor)
        un = RQ._static_unbatching_padding_asymmetric(b, int_bits, eb, factor)
        bs = int_bits // (eb + factor)
        assert [int(v) for v in un[:n]] == vals and len(un) == len(b) * bs
        out["batch"].append({"int_bits": int_bits, "element_bits": eb, "num_clients": C, "factor": factor, "n": n,
                             "vals": hxl(vals), "batched": hxl(b), "unbatched": hxl(un)})
    for eb, C, n, top_bits in [(32, 2, 100, 34), (16, 10, 64, 20), (32, 100, 50, 39), (16, 3, 40, 120), (8, 2, 30, 64)]:
        alpha = ACIQ(eb).get_alpha_gaus_direct(1.0)
        vals = rand_ints(rng, n, top_bits)
        r = RQ._static_unquantize_padding_asymmetric(np.array(vals, dtype=object), alpha, eb, C)
        out["unquantize"].append({"element_bits": eb, "num_clients": C, "n": n, "alpha": float(alpha).hex(),
                                  "vals": hxl(vals), "out": np.array([float(v) for v in r], dtype=np.float64).tobytes().hex()})
    dump("codec.json", out)


# --------------------------------------------------------------------------- sparsifier (f-3)
def gen_sparsify():
    """Client.sparsify (jzf_aggregator.py:578-623) CALLED on a stub client for one-layer models: top-k by |layer| (taken BEFORE
    the residual is added), residual accumulation across two rounds.  Its last three statements encode the location list with
    jzf_weights._to_bytes, which overflows under NumPy >= 2 (`s <<= ...` on a NumPy integer, jzf_weights.py:81 -- the wire format is
    covered by pack.json through _to_bytes_old): for the call, the module's _to_bytes / _from_bytes names are replaced by recorders,
    which is also how the location list -- a local of the method -- is read out.  Values are distinct in magnitude so NumPy's unstable
    argsort has a unique answer."""
    RA = _import_aggregator()
    seen = []

    def rec_to_bytes(locations, bits):
        seen.append(([int(v) for v in locations], int(bits)))
        return b"", 0

    RA._to_bytes, RA._from_bytes = rec_to_bytes, (lambda enc, le, bits: [])
    cases = []
    for dtype, n, sparsity, seed in [("float32", 1000, 0.01, 1), ("float32", 4099, 0.1, 2), ("float64", 777, 0.05, 3),
                                     ("float32", 50, 0.001, 4), ("float32", 300, 1.0, 5)]:
        rng = np.random.RandomState(seed)
        client = types.SimpleNamespace(remain_weights=None, _sparsity=sparsity, shape_dict_used_for_sparsification=None)
        rounds = []
        for rd in range(2):
            layer = rng.standard_normal(n).astype(dtype)
            while len(np.unique(np.abs(layer))) != n:          # float32 magnitudes can collide: redraw
                layer = rng.standard_normal(n).astype(dtype)
            w = _Weights({"layer": layer.copy()})
            enc, le, bits, base = RA.Client.sparsify(client, w)
            location, got_bits = seen.pop()
            assert base == n and bits == got_bits == int(n).bit_length() and not seen
            masked_layer, remain = w._weights["layer"], client.remain_weights["layer"]
            assert masked_layer.dtype == remain.dtype == np.dtype(dtype) and client.shape_dict_used_for_sparsification == {"layer": (n,)}
            rounds.append({"layer": layer.tobytes().hex(), "k": len(location), "location": location,
                           "masked": masked_layer.tobytes().hex(), "remain": remain.tobytes().hex()})
        cases.append({"dtype": dtype, "n": n, "sparsity": sparsity, "rounds": rounds})
    dump("sparsify.json", {"cases": cases})



# --------------------------------------------------------------------------- quantiser orchestration (f-1)
class _Weights:
    """What QuantizingClient walks (JZFOrderDictWeights surface it touches: walking_order, _weights)."""

    def __init__(self, layers):
        self.walking_order = sorted(layers)
        self._weights = dict(layers)


def _fhex(a, dtype=None):
    a = np.asarray(a if dtype is None else np.asarray(a).astype(dtype))
    return a.tobytes().hex()


def gen_quantclient():
    """ACIQ alpha (jzf_aciq.py:10-27) and QuantizingClient.normalize / quantize / unquantize / unnormalize
    (jzf_quantize.py:394-564, secure + padding path) driven on a stub weights object, two rounds, so that the per-layer
    mean / std carried from round to round (-> next round's alpha) is pinned too.  Batched and un-batched."""
    out = {"aciq": [], "clients": []}
    for bits in (2, 4, 8, 16, 20, 31, 32, 40):
        for sigma in (1.0, 0.037, 12.5):
            a = ACIQ(bits)
            out["aciq"].append({"bits": bits, "sigma": float(sigma).hex(), "alpha_direct": float(a.get_alpha_gaus_direct(sigma)).hex(),
                                "min": float(-3.2 * sigma).hex(), "max": float(2.9 * sigma).hex(), "size": 61706,
                                "alpha_gaus": float(a.get_alpha_gaus(-3.2 * sigma, 2.9 * sigma, 61706)).hex()})
    for int_bits, eb, C, batch, dtype, sizes, seed in [(64, 32, 2, False, "float32", [(5, 7), (11,)], 1), (128, 16, 10, True, "float32", [(33,), (4, 6), (2,)], 2),
                                                       (120, 16, 10, True, "float64", [(29,), (40,)], 3), (64, 16, 3, False, "float64", [(50,)], 4)]:
        rng = np.random.RandomState(seed)
        qc = RQ.QuantizingClient(int_bits, None, None, batch, eb, True, True)
        qc.num_clients = C
        rounds = []
        for rd in range(2):
            layers = {f"l{i}": (rng.standard_normal(sh) * (0.5 + i) + 0.1 * rd).astype(dtype) for i, sh in enumerate(sizes)}
            w = _Weights({k: v.copy() for k, v in layers.items()})
            if rd == 0:
                qc.set_layer_size_list(w)
            past_mean, past_std = list(qc.past_layer_mean_list), list(qc.past_layer_std_list)
            w = qc.normalize(w)
            normalized = {k: w._weights[k].copy() for k in w.walking_order}
            np.random.seed(500 + seed + rd)
            uniforms = {}
            st = np.random.get_state()
            for k in w.walking_order:                                  # the draws quantize() is about to make, layer by layer
                uniforms[k] = np.random.random(normalized[k].size)
            np.random.set_state(st)
            w = qc.quantize(w)
            quant = {k: [int(v) for v in np.asarray(w._weights[k]).flatten()] for k in w.walking_order}
            # the arbiter's sum of C uploads: this client's values plus C - 1 deterministic companions in the same range
            agg = {}
            for k in w.walking_order:
                top = (1 << eb) - 1
                if not batch:
                    other = [sum(int(rng.randint(0, top + 1)) for _ in range(C - 1)) for _ in quant[k]]
                    agg[k] = np.array([a + o for a, o in zip(quant[k], other)], dtype=object).reshape(layers[k].shape)
                else:
                    factor = int(np.ceil(np.log2(C)))
                    bs = int_bits // (eb + factor)
                    other = []
                    for _ in quant[k]:
                        t = 0
                        for _i in range(bs):
                            t = (t << (eb + factor)) + sum(int(rng.randint(0, top + 1)) for _ in range(C - 1))
                        other.append(t)
                    agg[k] = np.array([a + o for a, o in zip(quant[k], other)], dtype=object)
            w2 = _Weights({k: v.copy() for k, v in agg.items()})
            w2 = qc.unquantize(w2)
            unq = {k: np.array([float(v) for v in np.asarray(w2._weights[k]).flatten()], dtype=np.float64) for k in w2.walking_order}
            w2 = qc.unnormalize(w2)
            unn = {k: np.array([float(v) for v in np.asarray(w2._weights[k]).flatten()], dtype=np.float64) for k in w2.walking_order}
            rounds.append({"layers": {k: _fhex(v) for k, v in layers.items()}, "past_mean": [float(v).hex() for v in past_mean],
                           "past_std": [float(v).hex() for v in past_std], "normalized": {k: _fhex(v) for k, v in normalized.items()},
                           "uniforms": {k: _fhex(v) for k, v in uniforms.items()}, "alpha": [float(a).hex() for a in qc.alpha_list],
                           "quantized": {k: hxl(v) for k, v in quant.items()}, "aggregate": {k: hxl(np.asarray(v).flatten()) for k, v in agg.items()},
                           "unquantized": {k: _fhex(v) for k, v in unq.items()}, "unnormalized": {k: _fhex(v) for k, v in unn.items()},
                           "new_mean": [float(v).hex() for v in qc.past_layer_mean_list], "new_std": [float(v).hex() for v in qc.past_layer_std_list]})
        out["clients"].append({"int_bits": int_bits, "element_bits": eb, "num_clients": C, "batch": batch, "dtype": dtype,
                               "shapes": {f"l{i}": list(sh) for i, sh in enumerate(sizes)}, "rounds": rounds})
    dump("quantclient.json", out)


# --------------------------------------------------------------------------- adapter (a-16, f-4)
def _import_block():
    """jzf_flashe_block imports the Diffie-Hellman key exchange, which imports gmpy2 (absent here).  None of the code
    this generator runs touches it, so the module gets a stub whose names resolve."""
    gm = types.ModuleType("gmpy2")
    for nm in ("mpz", "powmod", "invert", "is_prime", "mpz_random", "random_state", "mpz_urandomb", "gcd", "lcm", "bit_length",
               "next_prime", "iroot", "c_mod", "t_mod", "f_mod", "divm", "isqrt", "mul", "add", "sub"):
        setattr(gm, nm, lambda *a, **k: None)
    sys.modules.setdefault("gmpy2", gm)
    from federatedml.framework.homo.procedure import jzf_flashe_block as RB
    return RB


def _import_aggregator():
    """federatedml/framework/homo/procedure/jzf_aggregator.py imports seven sibling block modules (:15-21); two are not in the tree
    (jzf_additive_mask_block, jzf_simple_block) and three need libraries this image lacks (bfv / ckks / paillier).  None of them is
    touched by the two methods called here, so they get empty stand-ins; jzf_flashe_block and jzf_plain_block import for real."""
    _import_block()
    pkg = "federatedml.framework.homo.procedure."
    for nm in ("jzf_additive_mask_block", "jzf_simple_block", "jzf_bfv_block", "jzf_ckks_block", "jzf_paillier_block"):
        sys.modules.setdefault(pkg + nm, types.ModuleType(pkg + nm))
    from federatedml.framework.homo.procedure import jzf_aggregator as RA
    return RA


def ref_expand_to_dense(RA, uploads, masks, total):
    """Arbiter.expand_to_dense (jzf_aggregator.py:150-165) itself, on stand-ins for the model objects it walks (one flattened layer
    per client: the compact ciphertexts followed by the un-encrypted quantised zero); returns the dense object arrays it leaves in
    mo._weights."""
    models = [types.SimpleNamespace(_weights={"flat": np.array(u, dtype=object)}) for u in uploads]
    RA.Arbiter.expand_to_dense(types.SimpleNamespace(), models, [list(m) for m in masks], total)
    return [m._weights["flat"] for m in models]


class _Wire:
    """Stand-in for a federation transfer variable: records what is sent, replays it on get()."""

    def __init__(self):
        self.sent = []

    def remote(self, obj=None, **kw):
        self.sent.append(obj)

    def get(self, **kw):
        return self.sent[-1]


def gen_block():
    """Arbiter.dynamic_masking's decision (jzf_flashe_block.py:89-117) and the _Client / Host forwarders (:120-174, :278-285)
    executed from the reference module on stub objects (the transport is a recorder): (1) cost-model cases; (2) a dense
    double-mask job with precompute over two rounds, the second with a dropout; (3) a sparse job with mask = "dynamic",
    where the arbiter's hint switches the clients to single masks over compact positions."""
    RB = _import_block()
    RA = _import_aggregator()
    rng = np.random.RandomState(321)
    dm = []
    for total, sets in [(40, [[1, 2, 3], [1, 2, 3], [1, 2, 3]]), (40, [[0, 5], [7, 9], [11]]), (30, [list(range(30))] * 4), (25, [[3, 4, 5, 6]]),
                        (50, [sorted(rng.choice(50, 20, replace=False).tolist()) for _ in range(5)]), (10, [[], []]),
                        (64, [sorted(rng.choice(64, 60, replace=False).tolist()) for _ in range(3)])]:
        g, h = _Wire(), _Wire()
        stub = types.SimpleNamespace(mask="dynamic", arbiter_to_guest=g, arbiter_to_host=h)
        RB.Arbiter.dynamic_masking(stub, [list(m) for m in sets], total, (0,))
        assert g.sent[0]["choice"] == h.sent[0]["choice"] and g.sent[0]["masks"] == [list(m) for m in sets]
        dm.append({"total": total, "masks": [list(m) for m in sets], "choice": g.sent[0]["choice"]})
    g = _Wire()
    assert RB.Arbiter.dynamic_masking(types.SimpleNamespace(mask="double", arbiter_to_guest=g, arbiter_to_host=g), [[1]], 5, (0,)) is None and not g.sent

    def client(b, idx, C, n, mask="double", precompute=True):
        """What Host.create_cipher leaves behind (:287-326), minus uuid sync and key exchange."""
        ci = RF.FlasheCipher(b)
        ci.idx = idx
        ci.generate_prp_seed(KEY)
        st = types.SimpleNamespace(cipher=ci, precompute=precompute, mask=mask, num_params=n,
                                   quantizer=types.SimpleNamespace(set_iter=lambda it: None), arbiter_to_host=None)
        if precompute:
            ci.set_num_params(n)
            ci.prepare_encrypt()
        ci.set_num_clients(C)
        return st

    dense = []
    for b, n, n_jobs, C in [(128, 37, 4, 3), (23, 50, 16, 4)]:
        RF.N_JOBS = n_jobs
        mod = 1 << b
        cls = [client(b, c, C, n) for c in range(C)]
        rounds = []
        for it, up in [(0, list(range(C))), (1, [c for c in range(C) if c != 1])]:
            pts, cts = {}, {}
            for c in up:
                RB._Client.set_iter_index(cls[c], it)
                pts[c] = rand_ints(rng, n, min(b, 64) - 4)
                had_masks = "add" in cls[c].cipher.next_iter_encrypt_prepared
                cts[c] = [int(v) for v in RB._Client.encrypt(cls[c], np.array(pts[c], dtype=object))]
                assert had_masks == (it == 0 or c in rounds[-1]["uploaded"])
            agg = reduce(lambda x, y: (x + y) % mod, [np.array(cts[c], dtype=object) for c in up])
            dec = {}
            for c in up:
                RB._Client.prepare_decrypt(cls[c])
                RB._Client.set_idx_list(cls[c], list(up))
                dec[c] = [int(v) for v in RB._Client.decrypt(cls[c], agg)]
                RB._Client.prepare_encrypt(cls[c])                    # idle-time precompute for the next round
            rounds.append({"iter": it, "uploaded": up, "pt": {str(c): hxl(pts[c]) for c in up}, "ct": {str(c): hxl(cts[c]) for c in up},
                           "agg": hxl(agg), "dec": {str(c): hxl(dec[c]) for c in up},
                           "roundtrip": dec[up[0]] == [sum(pts[c][j] for c in up) % mod for j in range(n)]})
        dense.append({"b": b, "n": n, "n_jobs": n_jobs, "num_clients": C, "rounds": rounds})
    sparse = []
    for b, total, n_jobs, C, k in [(128, 90, 4, 3, 12), (64, 70, 8, 4, 9)]:
        RF.N_JOBS = n_jobs
        mod = 1 << b
        it = 3
        masks = [sorted(rng.choice(total, k, replace=False).tolist()) for _ in range(C)]
        g, h = _Wire(), _Wire()
        RB.Arbiter.dynamic_masking(types.SimpleNamespace(mask="dynamic", arbiter_to_guest=g, arbiter_to_host=h), masks, total, (it,))
        cls = [client(b, c, C, k, mask="dynamic", precompute=False) for c in range(C)]
        pts, ups, zeros = [], [], []
        for c in range(C):
            cls[c].arbiter_to_host = h
            RB._Client.set_iter_index(cls[c], it)
            RB.Host.dynamic_masking(cls[c], (it,))                  # cipher.masking_scheme / cipher.masks from the hint
            cls[c].cipher.total = total                              # jzf_aggregator.py:707
            pt = rand_ints(rng, k, min(b, 64) - 4)
            pts.append(pt)
            zero = int(rng.randint(0, 1 << 15))
            zeros.append(zero)
            ups.append([int(v) for v in RB._Client.encrypt(cls[c], np.array(pt, dtype=object))] + [zero])
        dense_vecs = ref_expand_to_dense(RA, ups, masks, total)     # Arbiter.expand_to_dense (jzf_aggregator.py:150-165), called
        agg = reduce(lambda x, y: (x + y) % mod, dense_vecs)
        RB._Client.set_idx_list(cls[0], list(range(C)))
        dec = [int(v) for v in RB._Client.decrypt(cls[0], agg)]
        sparse.append({"b": b, "total": total, "n_jobs": n_jobs, "num_clients": C, "iter": it, "masks": masks, "choice": h.sent[0]["choice"],
                       "scheme_after_hint": cls[0].cipher.masking_scheme, "pt": [hxl(p) for p in pts], "zeros": zeros,
                       "uploads": [hxl(u) for u in ups], "agg": hxl(agg), "dec": hxl(dec)})
    dump("block.json", {"key": KEY.hex(), "dynamic_masking": dm, "dense_precompute": dense, "sparse_dynamic": sparse})


# --------------------------------------------------------------------------- the client step of a job
def gen_clientstep():
    """What a reference JOB does either side of the cipher, executed from the reference's own methods
    (Client.secure_aggregate, jzf_aggregator.py:721-743, and Client.aggregate's tail, :881-898):
        QuantizingClient.quantize -> Client.flatten_weights (:625-650) -> [sparse: strip the trailing quantised zero] ->
        JZFOrderDictWeights.encrypted(cipher=_Client) -> [re-append it]
    and back
        _Client.set_idx_list -> JZFOrderDictWeights.decrypted -> Client.unflatten_weights (:652-671) -> QuantizingClient.unquantize.
    flatten_weights leaves ONE vector, so the PRF counters -- and for int_bits <= 64 the chunks_idx chunking -- run across all layers.
    The aggregator-side methods are called unbound on stub objects (they only touch self.shape_dict); the weights are real
    JZFOrderDictWeights; the cipher is the reference FlasheCipher behind the reference _Client forwarders."""
    RB = _import_block()
    RA = _import_aggregator()
    rng = np.random.RandomState(77)
    cases = []
    #        b   eb  C  n_jobs scheme    iter layers (name, shape, dtype, scale)
    grid = [(128, 16, 3, 4, "double", 2, [("a_conv", (7, 5), "float32", 0.8), ("b_bias", (11,), "float32", 3.0), ("c_dense", (3, 4), "float64", 1.0)]),
            (64, 12, 3, 7, "double", 0, [("a_conv", (9, 4), "float32", 1.0), ("b_bias", (5,), "float32", 0.5), ("c_dense", (17,), "float32", 2.0)]),
            (20, 12, 3, 5, "double", 6, [("l0", (13, 3), "float32", 1.0), ("l1", (1,), "float32", 1.0), ("l2", (4, 5), "float64", 4.0)]),
            (64, 32, 2, 8, "single", 1, [("w", (10, 10), "float32", 1.0), ("x", (3,), "float32", 1.0)]),
            (23, 16, 4, 16, "double", 3, [("only", (61,), "float32", 1.0)])]
    # BATCHED jobs (the paper's main configuration, "batch": true): several quantised values per ciphertext element
    # (_static_batching_padding_asymmetric, jzf_quantize.py:162-185, every layer padded to whole elements on its own), then flattened
    batched = [(128, 16, 10, 8, "double", 4, [("a_conv", (7, 5), "float32", 0.8), ("b_bias", (11,), "float32", 3.0), ("c_dense", (3, 4), "float64", 1.0)]),
               (120, 16, 3, 16, "double", 1, [("l0", (13,), "float32", 1.0), ("l1", (6, 1), "float32", 2.0)]),
               (64, 12, 3, 5, "double", 7, [("w", (10, 10), "float32", 1.0), ("x", (3,), "float64", 1.0), ("y", (4,), "float32", 1.0)])]
    for spec in [g + (False,) for g in grid] + [g + (True,) for g in batched]:
        b, eb, C, n_jobs, scheme, it, layer_specs, batch = spec
        RF.N_JOBS = n_jobs
        mod = 1 << b
        clients = []
        for c in range(C):
            ci = RF.FlasheCipher(b, mask=scheme)
            ci.idx = c
            ci.generate_prp_seed(KEY)
            ci.set_num_clients(C)
            qc = RQ.QuantizingClient(b, None, None, batch, eb, True, True)
            qc.num_clients = C
            st = types.SimpleNamespace(cipher=ci, quantizer=qc, precompute=False, mask=scheme)
            st.encrypt = lambda x, st=st: RB._Client.encrypt(st, x)            # what JZFWeights.encrypted / decrypted call on the "cipher"
            st.decrypt = lambda x, st=st: RB._Client.decrypt(st, x)
            clients.append((st, types.SimpleNamespace(shape_dict=None)))       # (flashe_block client, aggregator-side Client stub)
        rec_clients, flat_cts = [], []
        for c, (st, agg_client) in enumerate(clients):
            RB._Client.set_iter_index(st, it)
            layers = {nm: (rng.standard_normal(sh) * sc).astype(dt) for nm, sh, dt, sc in layer_specs}
            w = RW.JZFOrderDictWeights({k: v.copy() for k, v in layers.items()})
            st.quantizer.set_layer_size_list(w)
            seed = 9000 + 10 * b + c
            np.random.seed(seed)
            w = st.quantizer.quantize(w)                                           # jzf_aggregator.py:722
            w = RA.Client.flatten_weights(agg_client, w)                           # :723
            assert len(w.walking_order) == 1
            flat_q = [int(v) for v in w._weights[w.walking_order[0]]]
            w = w.encrypted(cipher=st, inplace=True)                               # :741
            flat_ct = [int(v) for v in w._weights[w.walking_order[0]]]
            flat_cts.append(flat_ct)
            rec_clients.append({"seed": seed, "layers": {k: _fhex(v) for k, v in layers.items()}, "alpha": [float(a).hex() for a in st.quantizer.alpha_list],
                                "flat_key": str(w.walking_order[0]), "flat_quantized": hxl(flat_q), "flat_ct": hxl(flat_ct),
                                "shape_dict": {k: list(v) for k, v in agg_client.shape_dict.items()}})
        n = len(flat_cts[0])
        agg_elem = reduce(lambda x, y: (x + y) % mod, [np.array(v, dtype=object) for v in flat_cts])     # :424-430
        pmod = 1 << (b * n)
        agg_packed_int = reduce(lambda x, y: (x + y) % pmod, [packed_int(v, b) for v in flat_cts])      # :406-419
        agg_packed = RW._from_bytes_old(agg_packed_int, n, b)
        agg_packed.reverse()
        outs = {}
        for nm, agg in (("elem", agg_elem), ("packed", agg_packed)):
            st, agg_client = clients[0]
            RB._Client.set_idx_list(st, list(range(C)))                            # :883
            w = RW.JZFOrderDictWeights({rec_clients[0]["flat_key"]: np.array([int(v) for v in agg], dtype=object)})
            w = w.decrypted(cipher=st, inplace=True)                               # :887
            dec = [int(v) for v in w._weights[w.walking_order[0]]]
            w = RA.Client.unflatten_weights(agg_client, w)                         # :896
            w = RB._Client.unquantize(st, w)                                       # :899
            outs[nm] = {"dec": hxl(dec), "unquantized": {k: _fhex(np.array([float(v) for v in np.asarray(w._weights[k]).flatten()], dtype=np.float64))
                                                         for k in agg_client.shape_dict}}
        cases.append({"b": b, "element_bits": eb, "num_clients": C, "n_jobs": n_jobs, "scheme": scheme, "iter": it, "n": n, "batch": batch,
                      "layers": [[nm, list(sh), dt] for nm, sh, dt, _sc in layer_specs], "clients": rec_clients,
                      "agg_elem": hxl(agg_elem), "agg_packed": hxl(agg_packed), "out_elem": outs["elem"], "out_packed": outs["packed"]})

    # the sparse job: compact layers + the 'zzz' layer whose quantised zero rides un-encrypted behind the ciphertexts (:717-720, :735-743)
    sparse = []
    for b, eb, C, n_jobs, it, dense_specs, ks in [(128, 16, 3, 4, 5, [("a", (6, 5), "float32"), ("b", (20,), "float32")], [4, 3]),
                                                  (64, 12, 2, 8, 2, [("p", (40,), "float64"), ("q", (3, 7), "float32")], [6, 2])]:
        RF.N_JOBS = n_jobs
        mod = 1 << b
        total = int(sum(int(np.prod(sh)) for _nm, sh, _dt in dense_specs))
        masks, ups, recs, clients = [], [], [], []
        for c in range(C):
            base, loc = 0, []
            for (_nm, sh, _dt), k in zip(dense_specs, ks):
                size = int(np.prod(sh))
                loc += sorted((rng.choice(size, k, replace=False) + base).tolist())
                base += size
            masks.append(loc)
        g, h = _Wire(), _Wire()
        RB.Arbiter.dynamic_masking(types.SimpleNamespace(mask="dynamic", arbiter_to_guest=g, arbiter_to_host=h), masks, total, (it,))
        for c in range(C):
            ci = RF.FlasheCipher(b)
            ci.idx = c
            ci.generate_prp_seed(KEY)
            ci.set_num_clients(C)
            qc = RQ.QuantizingClient(b, None, None, False, eb, True, True)
            qc.num_clients = C
            st = types.SimpleNamespace(cipher=ci, quantizer=qc, precompute=False, mask="dynamic", arbiter_to_host=h)
            st.encrypt = lambda x, st=st: RB._Client.encrypt(st, x)
            st.decrypt = lambda x, st=st: RB._Client.decrypt(st, x)
            agg_client = types.SimpleNamespace(shape_dict=None, shape_dict_used_for_sparsification={nm: tuple(sh) for nm, sh, _dt in dense_specs})
            clients.append((st, agg_client))
            RB._Client.set_iter_index(st, it)
            ci.total = total                                                       # :707
            RB.Host.dynamic_masking(st, (it,))                                     # :709
            compact = {nm: (rng.standard_normal(k) * 1.5).astype(dt) for (nm, _sh, dt), k in zip(dense_specs, ks)}   # what sparsify leaves (:598)
            w = RW.JZFOrderDictWeights({k_: v.copy() for k_, v in compact.items()})
            st.quantizer.set_layer_size_list(w)                                    # (normalize's first call, before 'zzz' exists)
            w._weights['zzz'] = np.array([0.0])                                    # :717-718
            w.refresh_walking_order()
            seed = 7000 + 10 * b + c
            np.random.seed(seed)
            w = st.quantizer.quantize(w)
            w = RA.Client.flatten_weights(agg_client, w)
            k0 = w.walking_order[0]
            flat_q = [int(v) for v in w._weights[k0]]
            zero_quantized = w._weights[k0][-1]                                    # :735-737
            w._weights[k0] = w._weights[k0][:-1]
            w = w.encrypted(cipher=st, inplace=True)
            w._weights[k0] = np.append(w._weights[k0], [zero_quantized])           # :742-743
            up = [int(v) for v in w._weights[k0]]
            ups.append(up)
            recs.append({"seed": seed, "layers": {k_: _fhex(v) for k_, v in compact.items()}, "alpha": [float(a).hex() for a in st.quantizer.alpha_list],
                         "flat_key": str(k0), "flat_quantized": hxl(flat_q), "upload": hxl(up)})
        dense_vecs = ref_expand_to_dense(RA, ups, masks, total)
        agg = reduce(lambda x, y: (x + y) % mod, dense_vecs)
        st, agg_client = clients[0]
        RB._Client.set_idx_list(st, list(range(C)))
        w = RW.JZFOrderDictWeights({recs[0]["flat_key"]: np.array([int(v) for v in agg], dtype=object)})
        w = w.decrypted(cipher=st, inplace=True)
        dec = [int(v) for v in w._weights[w.walking_order[0]]]
        agg_client.shape_dict = agg_client.shape_dict_used_for_sparsification      # :893-894
        # (the sparse job unquantises DENSE layers with the alphas of the compact ones: alpha_list has one entry per layer either way)
        w = RA.Client.unflatten_weights(agg_client, w)
        w = RB._Client.unquantize(st, w)
        unq = {k_: _fhex(np.array([float(v) for v in np.asarray(w._weights[k_]).flatten()], dtype=np.float64)) for k_ in agg_client.shape_dict}
        sparse.append({"b": b, "element_bits": eb, "num_clients": C, "n_jobs": n_jobs, "iter": it, "total": total, "choice": h.sent[0]["choice"],
                       "dense_layers": [[nm, list(sh), dt] for nm, sh, dt in dense_specs], "ks": ks, "masks": masks, "clients": recs,
                       "agg": hxl(agg), "dec": hxl(dec), "unquantized": unq})
    dump("clientstep.json", {"key": KEY.hex(), "dense": cases, "sparse": sparse})


if __name__ == "__main__":
    gen_clientstep()
    gen_quantclient()
    gen_block()
    gen_sparsify()
    gen_codec()
    gen_aes()
    gen_masks()
    gen_rounds()
    gen_precompute()
    gen_pack()
    gen_sparse()
    gen_config1()
