#!/usr/bin/env python3
"""Differential fuzzer for the entry points added in round 4, each compared with the CPU oracle / NumPy on random shapes:
  range    flashe_encrypt_batch_range_dev   (every client's encrypt on one element slice, with and without the slice of their sum)
  model    flashe_quantize_encrypt_model_dev / flashe_decrypt_unquantize_model_dev (fused codec over a flattened model: random layer tables,
           float32 / float64 layers, empty layers, element sub-ranges)
  prepared flashe_prepare_* / flashe_*_prepared_dev (ctx-resident mask precompute, dropout extras)
  agg32    flashe_aggregate_elem_u32_dev    (uint32 reduce, up to 140 operands, unaligned views)
  bounds   flashe_span_bounds_* + sparse aggregate / decrypt through the handle, recompute for new lists
  masking  flashe_dynamic_masking_cost_dev  (positions shared by consecutive clients)
  strided  flashe_packed_resolve_carry_strided_dev (carry-in from triples walked backwards)
  rdptrs   flashe_aggregate_decrypt_range_dev at b > 64 on operands in separate allocations (the pointer-table one-launch form)
  spenc    flashe_sparse_encrypt_aggregate_dev (the span reduce with the PRF inside): ragged and CLUSTERED lists (spans with thousands of
           entries next to empty ones), any client indices, one- and two-limb plaintexts, bounds handle or none, argument tables built once,
           followed by the sparse decrypt of the result
usage: fuzz_round4.py [cases per family] [seed] [families, comma separated]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import engine as E  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes((5 * i + 11) & 255 for i in range(32))


def L(b):
    return (b + 63) // 64


def pick_n(rng, cap=3_000_000):
    kind = rng.integers(0, 5)
    if kind == 0:
        return int(rng.integers(1, 300))
    if kind == 1:
        return int(rng.integers(300, 70_000))
    if kind == 2:
        return int(256 * rng.integers(1, 2000) + rng.integers(-2, 3))
    if kind == 3:
        return int(min(cap, rng.integers(70_000, 1_200_000)))
    return int(min(cap, 1024 * 256 * rng.integers(1, 6) + rng.integers(-1, 2)))


def fuzz_range(rng, case):
    b = int(rng.choice([128, 128, 100, 65, 64, 40, 33, 23, 20, 8]))
    C = int(rng.choice([1, 2, 3, 10, 11, 40]))
    n = pick_n(rng, 24_000_000 // C)
    first = int(rng.integers(0, n)) if rng.random() < 0.8 else 0
    count = int(rng.integers(0, n - first + 1)) if rng.random() < 0.8 else n - first
    scheme = "double" if rng.random() < 0.8 else "single"
    i0 = int(rng.integers(0, 100))
    idx = [i0 + c for c in range(C)] if rng.random() < 0.85 else [int(v) for v in rng.integers(0, 50, C)]
    it, n_jobs = int(rng.integers(0, 2 ** 32)), int(rng.choice([1, 3, 8, 16]))
    with_sum = rng.random() < 0.7
    eng = E.Engine(KEY, b, device=0)
    Lb = L(b)
    pts = [rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for _ in idx]
    dpt = [eng.upload(p[first:first + count]) if count else eng.alloc(16) for p in pts]
    dct = [eng.alloc_vec(max(count, 1)) for _ in idx]
    dsum = eng.alloc_vec(max(count, 1)) if with_sum else None
    eng.encrypt_batch_range_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, n_jobs, first, count, dpt, 1, dct, dsum)
    want = [orc.encrypt(KEY, it, i, scheme, n_jobs, b, p)[first:first + count] for i, p in zip(idx, pts)]
    for v in sorted(set([0, C - 1] + [int(x) for x in rng.integers(0, C, 2)])):
        assert np.array_equal(dct[v].download(np.uint64, count * Lb).reshape(count, Lb), want[v]), ("range/ct", case, b, C, n, first, count, scheme, v)
    if with_sum and count:
        assert np.array_equal(dsum.download(np.uint64, count * Lb).reshape(count, Lb), orc.aggregate_elem(want, b)), ("range/sum", case, b, C, n, first, count)
    return f"b={b} C={C} n={n} [{first}, +{count}) {scheme} sum={with_sum}"


def fuzz_model(rng, case):
    b = int(rng.choice([128, 100, 64, 40, 23, 20, 16]))
    eb = int(rng.integers(2, min(b, 33) - 1))
    C = int(rng.choice([1, 2, 10]))
    if eb + int(np.ceil(np.log2(C + 1))) > b:
        eb = b - 4
    n_layers = int(rng.integers(1, 9))
    sizes = [int(rng.choice([0, 1, 7, 300, 5_000, 70_001, 300_000])) for _ in range(n_layers)]
    if sum(sizes) == 0:
        sizes[0] = 17
    dts = [np.float32 if rng.random() < 0.7 else np.float64 for _ in sizes]
    alphas = [float(rng.choice([0.1, 1.0, 2.5, 8.17])) for _ in sizes]
    n = sum(sizes)
    scheme = "double" if rng.random() < 0.8 else "single"
    it, idx, n_jobs = int(rng.integers(0, 1000)), int(rng.integers(0, 50)), int(rng.choice([1, 5, 16]))
    eng = E.Engine(KEY, b, device=0)
    Lb = L(b)
    layers = [(rng.standard_normal(s) * 1.3).astype(dt) for s, dt in zip(sizes, dts)]
    u = rng.random(n)
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)
    dx = [eng.upload(x) if x.size else eng.alloc(16) for x in layers]
    du = eng.upload(u)
    table = [(int(starts[i]), dx[i].ptr, alphas[i], dts[i] == np.float64) for i in range(n_layers)]
    first = int(rng.integers(0, n)) if rng.random() < 0.5 else 0
    count = int(rng.integers(0, n - first + 1)) if rng.random() < 0.5 else n - first
    ct = eng.alloc_vec(max(count, 1))
    eng.quantize_encrypt_model_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, n_jobs, first, count, table, eb,
                                   du.ptr + 8 * first, ct)
    # expectation: NumPy's arithmetic layer by layer in the layer's own dtype (jzf_quantize.py:55-67), then the oracle's encrypt of the flat vector
    q = np.zeros(n, dtype=np.uint64)
    for i, x in enumerate(layers):
        if not x.size:
            continue
        a = x.dtype.type(alphas[i])
        v = np.clip(x, -a, a) + a
        v = v * x.dtype.type((1 << eb) - 1) / (x.dtype.type(2) * a)
        q[starts[i]:starts[i] + x.size] = np.floor(v + u[starts[i]:starts[i] + x.size]).astype(np.int64).astype(np.uint64)
    want = orc.encrypt(KEY, it, idx, scheme, n_jobs, b, q)[first:first + count]
    assert np.array_equal(ct.download(np.uint64, count * Lb).reshape(count, Lb), want), ("model/enc", case, b, eb, sizes, first, count, scheme)
    # the way back: decrypt + unquantise of an "aggregate" of C such vectors (here: C times the same ciphertext stream shape)
    if count:
        cts = [orc.encrypt(KEY, it, c, "double", n_jobs, b, q) for c in range(C)]
        agg = orc.aggregate_elem(cts, b)
        dec = orc.limbs_to_ints(orc.decrypt(KEY, it, [C], [0], n_jobs, b, agg))
        dagg = eng.upload(agg[first:first + count])
        dout = eng.alloc(8 * count)
        eng.decrypt_unquantize_model_dev(it, [C], [0], n, n_jobs, first, count, dagg, [(int(starts[i]), None, alphas[i], False) for i in range(n_layers)],
                                         eb, C, dout)
        got = dout.download(np.float64, count)
        wantf = np.zeros(n)
        for i, s in enumerate(sizes):
            if s:
                a = alphas[i] * C
                v = np.array(dec[starts[i]:starts[i] + s], dtype=np.float64)
                wantf[starts[i]:starts[i] + s] = v * (2 * a) / (((1 << eb) - 1) * C) - a
        assert got.tobytes() == wantf[first:first + count].tobytes(), ("model/dec", case, b, eb, sizes, first, count)
    return f"b={b} eb={eb} layers={sizes} [{first}, +{count}) {scheme}"


def fuzz_prepared(rng, case):
    b = int(rng.choice([128, 100, 64, 33, 23, 20]))
    n = pick_n(rng, 1_500_000)
    C = int(rng.integers(2, 9))
    it, idx, n_jobs = int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 50)), int(rng.choice([1, 4, 16]))
    eng = E.Engine(KEY, b, device=0)
    Lb = L(b)
    pt = rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64)
    dp, out = eng.upload(pt), eng.alloc_vec(n)
    eng.prepare_encrypt(it, idx, 1, n, n_jobs)
    eng.encrypt_prepared_dev(n, dp, 1, out)
    assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), orc.encrypt(KEY, it, idx, "double", n_jobs, b, pt)), ("prepared/enc", case, b, n)
    assert not eng.prepared_query(eng.PREPARED_ENCRYPT)[0]
    up = sorted(set(int(v) for v in rng.integers(0, C, C))) if rng.random() < 0.7 else list(range(C))
    agg = rng.integers(0, 2 ** 64, (n, Lb), dtype=np.uint64)
    if b % 64:
        agg[:, Lb - 1] &= np.uint64((1 << (b % 64)) - 1)
    add, minus = E.telescope(sorted(up))
    eng.prepare_decrypt(it, C, n, n_jobs)
    eng.decrypt_prepared_dev(it, [i for i in add if i != C], [i for i in minus if i != 0], n, n_jobs, eng.upload(agg), out)
    # (the reference applies the precomputed masks unconditionally: +term(C) -term(0) plus whatever the telescoped lists add beyond them)
    want = orc.decrypt(KEY, it, [C] + [i for i in add if i != C], [0] + [i for i in minus if i != 0], n_jobs, b, agg)
    assert np.array_equal(out.download(np.uint64, n * Lb).reshape(n, Lb), want), ("prepared/dec", case, b, n, up)
    return f"b={b} n={n} C={C} uploaded={up}"


def fuzz_agg32(rng, case):
    b = int(rng.choice([32, 31, 23, 20, 16, 8, 1]))
    C = int(rng.choice([1, 2, 3, 10, 64, 65, 140]))
    n = pick_n(rng, 6_000_000 // C + 1)
    eng = E.Engine(KEY, b, device=0)
    off = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0          # operands that start at an odd element: not 16-byte aligned
    arrs = [rng.integers(0, 2 ** b, n + off, dtype=np.uint64).astype(np.uint32) for _ in range(C)]
    bufs = [eng.upload(a) for a in arrs]
    out = eng.alloc(4 * (n + off) + 16)
    eng.aggregate_elem_u32_dev([bf.ptr + 4 * off for bf in bufs], n, out.ptr + 4 * off)
    want = (sum(a[off:].astype(np.uint64) for a in arrs) & np.uint64((1 << b) - 1)).astype(np.uint32)
    assert np.array_equal(out.download(np.uint32, n + off)[off:], want), ("agg32", case, b, C, n, off)
    return f"b={b} C={C} n={n} off={off}"


def fuzz_bounds(rng, case):
    b = int(rng.choice([128, 100, 65]))
    total = int(rng.choice([4096, 4097, 50_000, 300_001, 2_000_000]))
    C = int(rng.choice([1, 3, 50, 64, 65, 90]))
    k = int(min(total, rng.choice([1, 40, 500, 4000])))
    eng = E.Engine(KEY, b, device=0)
    it, n_jobs = int(rng.integers(0, 1000)), int(rng.choice([1, 16]))

    def lists():
        return [np.sort(rng.choice(total, k, replace=False)).astype(np.uint32) for _ in range(C)]
    locs = lists()
    vals = [rng.integers(0, 2 ** 63, (k, 2), dtype=np.uint64) for _ in range(C)]
    dl, dv = [eng.upload(l) for l in locs], [eng.upload(v) for v in vals]
    zeros = [int(rng.integers(0, 2 ** 31)) for _ in range(C)]
    a0, a1, d0, d1 = (eng.alloc_vec(total) for _ in range(4))
    bnd = eng.span_bounds(total, dl, [k] * C)
    for rnd in range(2):
        eng.sparse_aggregate_dev(total, dl, [k] * C, dv, zeros, a0, sorted_lists=True)
        eng.sparse_aggregate_dev(total, dl, [k] * C, dv, zeros, a1, bounds=bnd)
        assert np.array_equal(a0.download(np.uint64, 2 * total), a1.download(np.uint64, 2 * total)), ("bounds/agg", case, total, C, k, rnd)
        eng.sparse_decrypt_dev(it, dl, [k] * C, total, n_jobs, a0, d0, sorted_lists=True)
        eng.sparse_decrypt_dev(it, dl, [k] * C, total, n_jobs, a1, d1, bounds=bnd)
        assert np.array_equal(d0.download(np.uint64, 2 * total), d1.download(np.uint64, 2 * total)), ("bounds/dec", case, total, C, k, rnd)
        if rnd == 0:
            mask = orc.sparse_minus_mask(KEY, it, locs, total, n_jobs, b)
            assert np.array_equal(d1.download(np.uint64, 2 * total).reshape(total, 2), orc.combine(b, a0.download(np.uint64, 2 * total).reshape(total, 2), None, mask))
            locs = lists()
            dl = [eng.upload(l) for l in locs]
            bnd.recompute(dl, [k] * C)
    return f"b={b} total={total} C={C} k={k}"


def fuzz_masking(rng, case):
    total = int(rng.choice([64, 1000, 70_000, 3_000_000]))
    C = int(rng.choice([1, 2, 5, 64, 65, 66, 130]))
    eng = E.Engine(KEY, 128, device=0)
    ks = [int(rng.integers(0, min(total, 3000) + 1)) for _ in range(C)]
    if rng.random() < 0.3:                                   # heavy overlap: lists drawn from a small pool
        pool = np.sort(rng.choice(total, min(total, 4000), replace=False))
        masks = [np.sort(rng.choice(pool, min(k, len(pool)), replace=False)).astype(np.uint32) for k in ks]
    else:
        masks = [np.sort(rng.choice(total, k, replace=False)).astype(np.uint32) for k in ks]
    ks = [len(m) for m in masks]
    dev = [eng.upload(m) if len(m) else eng.alloc(16) for m in masks]
    single, double = eng.dynamic_masking_cost_dev(dev, ks)
    canceled = sum(int(np.intersect1d(masks[i], masks[i + 1], assume_unique=True).size) for i in range(C - 1))
    assert single == 2 * sum(ks) and double == 2 * single - 2 * canceled, ("masking", case, total, C, single, double, canceled)
    return f"total={total} C={C} entries={sum(ks)} canceled={canceled}"


def fuzz_strided(rng, case):
    """x plus the carry-in derived from the triples of the less significant slices, walked with stride -3 from the last triple."""
    eng = E.Engine(KEY, 128, device=0)
    W = int(rng.integers(2, 10))
    g = int(rng.integers(0, W - 1))                          # this rank; the ranks after it are less significant
    n_limbs = int(rng.choice([1, 2, 5, 1000]))
    top = rng.random() < 0.3
    total_bits = 64 * n_limbs - (int(rng.integers(1, 64)) if top else 0)
    infos = np.zeros((W, 3), dtype=np.uint64)
    for r in range(W):
        low = [0, 1, 2 ** 64 - 1, 2 ** 64 - 2][int(rng.integers(0, 4))] if rng.random() < 0.7 else int(rng.integers(0, 2 ** 64, dtype=np.uint64))
        infos[r, 0], infos[r, 1], infos[r, 2] = np.uint64(low), np.uint64(int(rng.random() < 0.6)), np.uint64(int(rng.integers(0, 4)))
    x = rng.integers(0, 2 ** 64, n_limbs, dtype=np.uint64)
    if rng.random() < 0.5:
        x[:] = np.uint64(2 ** 64 - 1)
    if total_bits % 64:
        x[-1] &= np.uint64((1 << (total_bits % 64)) - 1)
    dinf, dx = eng.upload(infos.reshape(-1)), eng.upload(x)
    below = W - 1 - g
    eng.packed_resolve_carry_strided_dev(n_limbs, total_bits, dinf.ptr + 8 * 3 * (W - 1), below, -3, dx)
    carry = 0
    for r in range(W - 1, g, -1):                            # lowest slice first
        low, ones, cout = (int(v) for v in infos[r])
        carry = cout + (1 if (ones and low + carry >= 1 << 64) else 0)
    want = (int.from_bytes(x.tobytes(), "little") + carry) % (1 << total_bits)
    got = int.from_bytes(dx.download(np.uint64, n_limbs).tobytes(), "little")
    assert got == want, ("strided", case, W, g, n_limbs, total_bits, hex(got)[:30], hex(want)[:30])
    return f"W={W} rank={g} limbs={n_limbs} bits={total_bits} carry={carry}"


def fuzz_rdptrs(rng, case):
    """flashe_aggregate_decrypt_range_dev at b > 64 with operands scattered over separate allocations: the pointer-table one-launch form
    (up to 64 operands; beyond, or with prefix lists, the two-launch form), sub-ranges, with / without the stored aggregate."""
    b = int(rng.choice([128, 128, 127, 100, 65]))
    C = int(rng.choice([1, 2, 3, 10, 11, 12, 33, 64, 65, 70]))
    n = pick_n(rng, 12_000_000 // C + 1)
    first = int(rng.integers(0, n)) if rng.random() < 0.6 else 0
    count = int(rng.integers(0, n - first + 1)) if rng.random() < 0.6 else n - first
    it, n_jobs = int(rng.integers(0, 2 ** 32)), int(rng.choice([1, 16]))
    kind = rng.integers(0, 4)
    add, minus = ([C], [0]) if kind < 2 else ([int(rng.integers(0, 99))], []) if kind == 2 else ([3, 7], [0, 5])
    eng = E.Engine(KEY, b, device=0)
    cts = []
    for _ in range(C):
        a = rng.integers(0, 2 ** 64, (n, 2), dtype=np.uint64)
        if b < 128:
            a[:, 1] &= np.uint64((1 << (b - 64)) - 1)
        cts.append(a)
    dev = [eng.upload(c) for c in cts]
    keep = rng.random() < 0.5
    out, ao = eng.alloc_vec(max(count, 1)), eng.alloc_vec(max(count, 1))
    eng.aggregate_decrypt_range_dev(it, add, minus, n, n_jobs, first, count, [d.ptr + 16 * first for d in dev], ao if keep else None, out)
    if count:
        agg = orc.aggregate_elem([c[first:first + count] for c in cts], b)
        want = orc.combine(b, agg, orc.mask_sum(KEY, it, add, n, n_jobs, b)[first:first + count], orc.mask_sum(KEY, it, minus, n, n_jobs, b)[first:first + count])
        assert np.array_equal(out.download(np.uint64, 2 * count).reshape(count, 2), want), ("rdptrs", case, b, C, n, first, count, add, minus)
        if keep:
            assert np.array_equal(ao.download(np.uint64, 2 * count).reshape(count, 2), agg), ("rdptrs/agg", case, b, C, n, first, count)
    return f"b={b} C={C} n={n} [{first}, +{count}) add={add} minus={minus} keep={keep}"


def fuzz_spenc(rng, case):
    b = int(rng.choice([128, 128, 100, 65, 64, 23]))
    lim = L(b)
    total = int(rng.choice([1_751, 1_753, 4_096, 50_000, 300_001, 1_500_000]))
    C = int(rng.choice([1, 2, 7, 50, 64, 65, 80]))
    pt_limbs = int(rng.choice([1, lim]))
    it, n_jobs = int(rng.integers(0, 1000)), int(rng.choice([1, 16]))
    ks, locs = [], []
    for c in range(C):
        style = rng.integers(0, 4)
        if style == 0:                                     # uniform
            kc = int(min(total, rng.choice([0, 1, 40, 500, 4000])))
            l = np.sort(rng.choice(total, kc, replace=False))
        elif style == 1:                                   # one dense cluster (a layer whose values all made the cut)
            kc = int(min(total, rng.choice([300, 3000, 20_000])))
            a = int(rng.integers(0, total - kc + 1))
            l = np.arange(a, a + kc)
        elif style == 2:                                   # a cluster plus a sparse tail
            kc1 = int(min(total // 2, rng.choice([100, 2500])))
            a = int(rng.integers(0, total // 2 - kc1 + 1))
            tail = np.sort(rng.choice(np.arange(total // 2, total), int(min(total - total // 2, rng.choice([0, 10, 700]))), replace=False))
            l = np.concatenate([np.arange(a, a + kc1), tail])
        else:                                              # every position
            l = np.arange(total) if total <= 60_000 else np.sort(rng.choice(total, 1000, replace=False))
        locs.append(l.astype(np.uint32)); ks.append(int(l.size))
    idx = [int(v) for v in rng.integers(0, 2 ** 20, C)]
    top = 2 ** 63 if b >= 64 else 2 ** (b - 1)
    pts = [rng.integers(0, top, (kc, pt_limbs), dtype=np.uint64) for kc in ks]
    if lim == 2 and pt_limbs == 2 and b < 128:
        for p_ in pts:
            p_[:, 1] &= np.uint64((1 << (b - 64)) - 1)
    zeros = [int(rng.integers(0, 2 ** min(b - 2, 31))) for _ in range(C)]
    eng = E.Engine(KEY, b, device=0)
    dl = [eng.upload(l) if l.size else eng.alloc(16) for l in locs]
    dp = [eng.upload(p_) if p_.size else eng.alloc(16) for p_ in pts]
    cts = [eng.alloc_vec(max(kc, 1)) for kc in ks]
    agg, dec = eng.alloc_vec(total), eng.alloc_vec(total)
    use_tables, use_bounds = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    a_loc, a_k, a_pt, a_ct = (eng.ptr_table(dl), eng.u64_table(ks), eng.ptr_table(dp), eng.ptr_table(cts)) if use_tables else (dl, ks, dp, cts)
    bnd = eng.span_bounds(total, a_loc, a_k) if use_bounds else None
    eng.sparse_encrypt_aggregate_dev(it, idx, a_loc, a_k, a_pt, pt_limbs, eng.zeros_table(zeros) if use_tables else zeros, total, n_jobs, a_ct, agg, bounds=bnd)
    want = np.zeros((total, lim), dtype=np.uint64)
    for c in range(C):
        wc = orc.encrypt(KEY, it, idx[c], "single", n_jobs, b, pts[c]) if ks[c] else np.zeros((0, lim), dtype=np.uint64)
        if ks[c]:
            assert np.array_equal(cts[c].download(np.uint64, ks[c] * lim).reshape(ks[c], lim), wc), ("spenc/ct", case, b, total, C, c)
        z = np.array([[zeros[c]] + [0] * (lim - 1)], dtype=np.uint64)
        want = orc.aggregate_elem([want, orc.expand_to_dense(total, locs[c], wc, z, b)], b)
    got = agg.download(np.uint64, total * lim).reshape(total, lim)
    assert np.array_equal(got, want), ("spenc/agg", case, b, total, C, ks[:6])
    if lim == 2 and len(set(idx)) == C:
        # the decrypt twin on the same lists (client c's prefix is c there: the minus-mask of lists 0 .. C-1)
        eng.sparse_decrypt_dev(it, a_loc, a_k, total, n_jobs, agg, dec, sorted_lists=True, bounds=bnd)
        mask = orc.sparse_minus_mask(KEY, it, locs, total, n_jobs, b)
        assert np.array_equal(dec.download(np.uint64, 2 * total).reshape(total, 2), orc.combine(b, got, None, mask)), ("spenc/dec", case, b, total, C)
    return f"b={b} total={total} C={C} pt_limbs={pt_limbs} tables={use_tables} bounds={use_bounds} k={ks[:5]}"


FAMILIES = {"spenc": fuzz_spenc, "rdptrs": fuzz_rdptrs, "range": fuzz_range, "model": fuzz_model, "prepared": fuzz_prepared, "agg32": fuzz_agg32, "bounds": fuzz_bounds, "masking": fuzz_masking,
            "strided": fuzz_strided}


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fams = sys.argv[3].split(",") if len(sys.argv) > 3 else list(FAMILIES)
    orc.build()
    total = 0
    for name in fams:
        rng = np.random.Generator(np.random.PCG64([seed, sum(map(ord, name))]))
        for case in range(per):
            desc = FAMILIES[name](rng, case)
            total += 1
            if os.environ.get("FUZZ_VERBOSE"):
                print(name, case, desc, flush=True)
    print(f"FUZZ_R4_OK {total} cases (seed {seed}, families {','.join(fams)})")


if __name__ == "__main__":
    main()
