"""ctypes wrapper around oracle/libflashe_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (flashe_amd/) never does.  See flashe_oracle.c for the
reference file:line each function restates.

Arrays are numpy uint64, shape [n, L] with L = 1 (b <= 64) or 2 (b > 64), little-endian
limbs.  `ints_to_limbs` / `limbs_to_ints` convert to and from Python ints.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libflashe_oracle.so")

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "flashe_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libflashe_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.fo_num_threads.restype = ctypes.c_int
    return _lib


def limbs_of(b):
    return 2 if b > 64 else 1


def ints_to_limbs(vals, b):
    L = limbs_of(b)
    out = np.zeros((len(vals), L), dtype=np.uint64)
    m64 = (1 << 64) - 1
    for j, v in enumerate(vals):
        v = int(v)
        out[j, 0] = v & m64
        if L == 2:
            out[j, 1] = (v >> 64) & m64
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    if arr.shape[1] == 1:
        return [int(x) for x in arr[:, 0]]
    return [int(lo) | (int(hi) << 64) for lo, hi in zip(arr[:, 0], arr[:, 1])]


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _key(key):
    key = bytes(key)
    assert len(key) == 32
    return (ctypes.c_uint8 * 32).from_buffer_copy(key)


def _idx(lst):
    a = np.ascontiguousarray(np.asarray(list(lst), dtype=np.uint32))
    return a, a.ctypes.data_as(_u32p), len(a)


def aes256_encrypt_block(key, block):
    rk = (ctypes.c_uint32 * 60)()
    lib().fo_aes256_key_expand(_key(key), rk)
    out = (ctypes.c_uint8 * 16)()
    lib().fo_aes256_encrypt_block(rk, (ctypes.c_uint8 * 16).from_buffer_copy(bytes(block)), out)
    return bytes(out)


def aes256_round_keys(key):
    rk = (ctypes.c_uint32 * 60)()
    lib().fo_aes256_key_expand(_key(key), rk)
    return list(rk)


def chunks(n, n_jobs):
    b = np.zeros(n_jobs + 1, dtype=np.uint64)
    lib().fo_chunks(ctypes.c_uint64(n), ctypes.c_uint32(n_jobs), _p64(b))
    return [int(x) for x in b]


def mask(key, it, idx, n, n_jobs, b):
    out = np.zeros((n, limbs_of(b)), dtype=np.uint64)
    lib().fo_mask(_key(key), ctypes.c_uint32(it), ctypes.c_uint32(idx), ctypes.c_uint64(n),
                  ctypes.c_uint32(n_jobs), ctypes.c_int(b), _p64(out))
    return out


def mask_sum(key, it, idx_list, n, n_jobs, b):
    out = np.zeros((n, limbs_of(b)), dtype=np.uint64)
    a, p, k = _idx(idx_list)
    lib().fo_mask_sum(_key(key), ctypes.c_uint32(it), p, ctypes.c_int(k), ctypes.c_uint64(n),
                      ctypes.c_uint32(n_jobs), ctypes.c_int(b), _p64(out))
    return out


def combine(b, inp, add=None, minus=None):
    inp = np.ascontiguousarray(inp, dtype=np.uint64)
    if inp.ndim == 1:
        inp = inp.reshape(-1, 1)
    n = inp.shape[0]
    L = limbs_of(b)
    out = np.zeros((n, L), dtype=np.uint64)
    pa = _p64(np.ascontiguousarray(add)) if add is not None else None
    pm = _p64(np.ascontiguousarray(minus)) if minus is not None else None
    lib().fo_combine(ctypes.c_uint64(n), ctypes.c_int(b), _p64(inp), ctypes.c_int(inp.shape[1]),
                     pa, pm, _p64(out))
    return out


def encrypt(key, it, idx, scheme, n_jobs, b, pt, out=None):
    """scheme: 'single' | 'double'.  pt: uint64 [n] / [n,1] (zero-extended) or [n,L].  out: optional
    preallocated [n, L] result buffer (the CPU baseline reuses pre-touched buffers)."""
    pt = np.ascontiguousarray(pt, dtype=np.uint64)
    if pt.ndim == 1:
        pt = pt.reshape(-1, 1)
    n = pt.shape[0]
    ct = np.zeros((n, limbs_of(b)), dtype=np.uint64) if out is None else out
    rc = lib().fo_encrypt(_key(key), ctypes.c_uint32(it), ctypes.c_uint32(idx),
                          ctypes.c_int(1 if scheme == "double" else 0), ctypes.c_uint64(n),
                          ctypes.c_uint32(n_jobs), ctypes.c_int(b), _p64(pt),
                          ctypes.c_int(pt.shape[1]), _p64(ct))
    assert rc == 0
    return ct


def decrypt(key, it, add_idx, minus_idx, n_jobs, b, ct, out=None):
    ct = np.ascontiguousarray(ct, dtype=np.uint64)
    if ct.ndim == 1:
        ct = ct.reshape(-1, 1)
    n = ct.shape[0]
    assert ct.shape[1] == limbs_of(b)
    out = np.zeros_like(ct) if out is None else out
    aa, pa, ka = _idx(add_idx)
    am, pm, km = _idx(minus_idx)
    rc = lib().fo_decrypt(_key(key), ctypes.c_uint32(it), pa, ctypes.c_int(ka), pm, ctypes.c_int(km),
                          ctypes.c_uint64(n), ctypes.c_uint32(n_jobs), ctypes.c_int(b), _p64(ct), _p64(out))
    assert rc == 0
    return out


def telescope(raw_idx_list):
    """-> (add_idx, minus_idx) as in FlasheCipher.set_idx_list(mode='decrypt')."""
    a, p, k = _idx(raw_idx_list)
    add = np.zeros(max(k, 1), dtype=np.uint32)
    minus = np.zeros(max(k, 1), dtype=np.uint32)
    r = lib().fo_telescope(p, ctypes.c_int(k), add.ctypes.data_as(_u32p), minus.ctypes.data_as(_u32p))
    return [int(x) for x in add[:r]], [int(x) for x in minus[:r]]


def _ptr_table(arrs):
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in arrs]
    tab = (_u64p * len(arrs))(*[_p64(a) for a in arrs])
    return arrs, tab


def aggregate_elem(cts, b, out=None):
    arrs, tab = _ptr_table(cts)
    n = arrs[0].shape[0]
    out = np.zeros((n, limbs_of(b)), dtype=np.uint64) if out is None else out
    lib().fo_aggregate_elem(ctypes.c_int(len(arrs)), tab, ctypes.c_uint64(n), ctypes.c_int(b), _p64(out))
    return out


def aggregate_packed(packed, total_bits):
    arrs, tab = _ptr_table(packed)
    n_limbs = (total_bits + 63) // 64
    out = np.zeros(n_limbs, dtype=np.uint64)
    lib().fo_aggregate_packed(ctypes.c_int(len(arrs)), tab, ctypes.c_uint64(n_limbs),
                              ctypes.c_uint64(total_bits), _p64(out))
    return out


def pack(x, b):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    n = x.shape[0]
    out = np.zeros((n * b + 63) // 64, dtype=np.uint64)
    lib().fo_pack(ctypes.c_uint64(n), ctypes.c_int(b), _p64(x), _p64(out))
    return out


def unpack(p, n, b):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = np.zeros((n, limbs_of(b)), dtype=np.uint64)
    lib().fo_unpack(ctypes.c_uint64(n), ctypes.c_int(b), _p64(p), _p64(out))
    return out


def expand_to_dense(total, loc, vals, zero, b):
    loc = np.ascontiguousarray(loc, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    zero = np.ascontiguousarray(zero, dtype=np.uint64).reshape(-1)
    out = np.zeros((total, limbs_of(b)), dtype=np.uint64)
    lib().fo_expand_to_dense(ctypes.c_uint64(total), ctypes.c_uint64(len(loc)),
                             loc.ctypes.data_as(_u32p), _p64(vals), _p64(zero), ctypes.c_int(b), _p64(out))
    return out


def sparse_minus_mask(key, it, locs, total, n_jobs, b):
    locs = [np.ascontiguousarray(l, dtype=np.uint32) for l in locs]
    tab = (_u32p * len(locs))(*[l.ctypes.data_as(_u32p) for l in locs])
    k = np.asarray([len(l) for l in locs], dtype=np.uint64)
    out = np.zeros((total, limbs_of(b)), dtype=np.uint64)
    rc = lib().fo_sparse_minus_mask(_key(key), ctypes.c_uint32(it), ctypes.c_int(len(locs)), tab, _p64(k),
                                    ctypes.c_uint64(total), ctypes.c_uint32(n_jobs), ctypes.c_int(b), _p64(out))
    assert rc == 0
    return out


def sparse_dense_mask(key, it, sels, total, b):
    sels = [np.ascontiguousarray(s, dtype=np.uint8) for s in sels]
    tab = (_u8p * len(sels))(*[s.ctypes.data_as(_u8p) for s in sels])
    out = np.zeros((total, limbs_of(b)), dtype=np.uint64)
    lib().fo_sparse_dense_mask(_key(key), ctypes.c_uint32(it), ctypes.c_int(len(sels)), tab,
                               ctypes.c_uint64(total), ctypes.c_int(b), _p64(out))
    return out


def quantize(x, alpha, bits, u):
    x = np.ascontiguousarray(x)
    u = np.ascontiguousarray(u, dtype=np.float64)
    q = np.zeros(len(x), dtype=np.uint64)
    if x.dtype == np.float32:
        lib().fo_quantize_f32(ctypes.c_uint64(len(x)), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(alpha),
                              ctypes.c_int(bits), u.ctypes.data_as(ctypes.c_void_p), _p64(q))
    else:
        x = x.astype(np.float64)
        lib().fo_quantize_f64(ctypes.c_uint64(len(x)), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(alpha),
                              ctypes.c_int(bits), u.ctypes.data_as(ctypes.c_void_p), _p64(q))
    return q


def unquantize(v, alpha, bits, num_clients):
    v = np.ascontiguousarray(v, dtype=np.uint64)
    if v.ndim == 1:
        v = v.reshape(-1, 1)
    out = np.zeros(v.shape[0], dtype=np.float64)
    lib().fo_unquantize(ctypes.c_uint64(v.shape[0]), _p64(v), ctypes.c_int(v.shape[1]), ctypes.c_double(alpha),
                        ctypes.c_int(bits), ctypes.c_int(num_clients), out.ctypes.data_as(ctypes.c_void_p))
    return out


def batch(vals, int_bits, field_bits):
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    bs = int_bits // field_bits
    nb = (len(vals) + bs - 1) // bs
    out = np.zeros((nb, limbs_of(int_bits)), dtype=np.uint64)
    lib().fo_batch.restype = ctypes.c_uint64
    r = lib().fo_batch(ctypes.c_uint64(len(vals)), _p64(vals), ctypes.c_int(int_bits), ctypes.c_int(field_bits), _p64(out))
    assert r == nb
    return out


def unbatch(batched, int_bits, field_bits):
    batched = np.ascontiguousarray(batched, dtype=np.uint64)
    if batched.ndim == 1:
        batched = batched.reshape(-1, 1)
    bs = int_bits // field_bits
    out = np.zeros(batched.shape[0] * bs, dtype=np.uint64)
    lib().fo_unbatch(ctypes.c_uint64(batched.shape[0]), _p64(batched), ctypes.c_int(int_bits), ctypes.c_int(field_bits), _p64(out))
    return out


def sparsify(layer, k, remain):
    """-> (loc uint32[k], vals[k], new remain).  remain is not modified in place."""
    layer = np.ascontiguousarray(layer)
    is64 = layer.dtype == np.float64
    if not is64:
        layer = layer.astype(np.float32)
    rem = np.ascontiguousarray(remain, dtype=layer.dtype).copy()
    loc = np.zeros(k, dtype=np.uint32)
    vals = np.zeros(k, dtype=layer.dtype)
    rc = lib().fo_sparsify(ctypes.c_uint64(len(layer)), ctypes.c_uint64(k), layer.ctypes.data_as(ctypes.c_void_p),
                           ctypes.c_int(1 if is64 else 0), rem.ctypes.data_as(ctypes.c_void_p),
                           loc.ctypes.data_as(_u32p), vals.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return loc, vals, rem


def aesni_available():
    return bool(lib().fo_aesni_available())


def set_aesni(on):
    """Force the portable table AES (False) or allow AES-NI (True)."""
    lib().fo_set_aesni(ctypes.c_int(1 if on else 0))


def vaes_available():
    return bool(lib().fo_vaes_available())


def set_vaes(on):
    """Allow (True) or forbid (False) the AVX-512 VAES form of the AES-NI path."""
    lib().fo_set_vaes(ctypes.c_int(1 if on else 0))


def num_threads():
    return int(lib().fo_num_threads())


def set_num_threads(t):
    lib().fo_set_num_threads(ctypes.c_int(t))
