#!/usr/bin/env python3
"""Differential fuzzer for what round 5 added, each family compared with the CPU oracle / NumPy on random shapes:
  fixed    the compact layout at the compile-time widths (int_bits 16 / 20 / 23 / 24 / 32; 8: the run-time width, paired) in launches long enough for the paired kernel: random
           lengths, chunkings (n_jobs 1 .. thousands: chunk ends everywhere), client runs, single / double mask, in place or not
  fixed64  int_bits 64 at compile time (one-limb layout), whole vectors and element sub-ranges that start or end inside a block
  encsum   flashe_encrypt_batch_sum_u32_dev: the one-launch form (consecutive clients, double mask, long vectors) and every shape that
           falls back to encrypts + reduce (short vectors, other widths, gaps in the client indices, single mask, > 128 clients)
  combsum  flashe_combine_batch_sum_dev / flashe_combine_batch_sum_decrypt_dev (round 6): 0 .. 150 vectors, missing add / minus entries, batches without any minus operand, one- and two-limb inputs, every modulus class
  sparsify flashe_sparsify_batch_dev / flashe_sparsify_dev (the rewritten streaming passes): random layer tables incl. empty and tiny
           layers, float32 / float64, ties at the threshold (quantised values), residuals; against a NumPy restatement of
           Client.sparsify's ranking (jzf_aggregator.py:578-623: |x| before the residual is added, ties to the higher index)
  idx32    the raw-ABI range check: idx = 2^32 - 1 refused under the double mask, accepted under the single mask
usage: fuzz_round5.py [cases per family] [seed] [families, comma separated]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import engine as E  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes((7 * i + 3) & 255 for i in range(32))


def L(b):
    return (b + 63) // 64


def fuzz_fixed(rng, case):
    b = int(rng.choice([20, 20, 23, 16, 8, 24, 32]))
    m = 128 // b
    # at least 2 * 128 * 4096 blocks per stream so that the paired (compile-time width) kernel runs; sometimes just below (generic kernel)
    n_min = 2 * 128 * 4096 * m
    n = int(n_min + rng.integers(-3 * m, 200_000)) if rng.random() < 0.85 else int(rng.integers(1, n_min))
    J = int(rng.choice([1, 2, 7, 16, 16, 100, 5000, 40_000]))
    C = int(rng.choice([1, 2, 3]))
    scheme = "double" if rng.random() < 0.75 else "single"
    i0 = int(rng.integers(0, 1000))
    idx = [i0 + c for c in range(C)] if rng.random() < 0.8 else sorted(int(v) for v in rng.choice(1000, C, replace=False))
    it = int(rng.integers(0, 2 ** 32))
    eng = E.Engine(KEY, b, device=0)
    pts = [rng.integers(0, 2 ** b, n, dtype=np.uint64) for _ in idx]
    off = int(rng.integers(0, 4))                                       # words past a 16-byte boundary
    dpt = [eng.alloc(4 * (n + 8)) for _ in idx]
    for d, p in zip(dpt, pts):
        d.upload_at(4 * off, p.astype(np.uint32))
    inplace = rng.random() < 0.3
    dct = dpt if inplace else [eng.alloc(4 * (n + 8)) for _ in idx]
    eng.encrypt_batch_u32_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, J,
                              [d.ptr + 4 * off for d in dpt], [d.ptr + 4 * off for d in dct])
    for v, i in enumerate(idx):
        want = orc.encrypt(KEY, it, i, scheme, J, b, pts[v])[:, 0].astype(np.uint32)
        got = dct[v].download(np.uint32, n + off)[off:]
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, ("fixed", case, b, n, J, idx, scheme, off, inplace, v, bad[:6])
    return f"b={b} n={n} J={J} idx={idx} {scheme} off={off} inplace={inplace}"


def fuzz_fixed64(rng, case):
    n_min = 2 * 128 * 4096 * 2
    n = int(n_min + rng.integers(-4, 150_000)) if rng.random() < 0.85 else int(rng.integers(1, n_min))
    J = int(rng.choice([1, 3, 16, 16, 999, 30_001]))
    C = int(rng.choice([1, 2, 3]))
    scheme = "double" if rng.random() < 0.8 else "single"
    idx = list(range(5, 5 + C))
    it = int(rng.integers(0, 2 ** 32))
    eng = E.Engine(KEY, 64, device=0)
    pts = [rng.integers(0, 2 ** 64, n, dtype=np.uint64) for _ in idx]
    first = int(rng.integers(0, min(n, 9))) if rng.random() < 0.6 else 0
    count = n - first - (int(rng.integers(0, min(n - first, 9))) if rng.random() < 0.6 else 0)
    dpt = [eng.upload(p[first:first + count]) if count else eng.alloc(16) for p in pts]
    dct = [eng.alloc_vec(max(count, 1)) for _ in idx]
    eng.encrypt_batch_range_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, J, first, count, dpt, 1, dct)
    for v, i in enumerate(idx):
        want = orc.encrypt(KEY, it, i, scheme, J, 64, pts[v])[first:first + count, 0]
        assert np.array_equal(dct[v].download(np.uint64, count), want), ("fixed64", case, n, J, first, count, scheme, v)
    return f"n={n} J={J} C={C} {scheme} range=({first},{count})"


def fuzz_encsum(rng, case):
    b = int(rng.choice([20, 20, 23, 16, 25, 8, 32]))
    m = 128 // b
    long_vec = rng.random() < 0.5
    n = int(2 * 128 * 4096 * m + rng.integers(0, 50_000)) if long_vec else int(rng.integers(1, 400_000))
    C = int(rng.choice([1, 2, 3, 5])) if long_vec else int(rng.choice([1, 2, 7, 40, 130]))
    J = int(rng.choice([1, 8, 16, 333]))
    scheme = "double" if rng.random() < 0.8 else "single"
    i0 = int(rng.integers(0, 50))
    idx = [i0 + c for c in range(C)]
    if rng.random() < 0.2 and C > 1:
        idx[-1] += 3                                                    # a gap: not one run of consecutive clients
    it = int(rng.integers(0, 2 ** 32))
    eng = E.Engine(KEY, b, device=0)
    base = [rng.integers(0, 2 ** b, n, dtype=np.uint64) for _ in range(min(C, 3))]
    dpt = [eng.upload(base[v % len(base)].astype(np.uint32)) for v in range(C)]
    dct = [eng.alloc(4 * n + 16) for _ in range(C)]
    dsum = eng.alloc(4 * n + 16)
    eng._check(eng._lib.flashe_memset_dev(eng._h, dsum.ptr, 0xE1, dsum.nbytes))
    eng.encrypt_batch_sum_u32_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, J, dpt, dct, dsum)
    wsum = np.zeros(n, dtype=np.uint64)
    for v, i in enumerate(idx):
        w = orc.encrypt(KEY, it, i, scheme, J, b, base[v % len(base)])[:, 0]
        wsum += w
        if v < 3 or v == C - 1:
            assert np.array_equal(dct[v].download(np.uint32, n).astype(np.uint64), w), ("encsum/ct", case, b, n, C, J, scheme, v)
    bad = np.flatnonzero(dsum.download(np.uint32, n).astype(np.uint64) != (wsum & np.uint64((1 << b) - 1)))
    assert bad.size == 0, ("encsum/sum", case, b, n, C, J, scheme, idx[:4], bad[:6])
    return f"b={b} n={n} C={C} J={J} {scheme} idx0={idx[0]} last={idx[-1]}"


def fuzz_combsum(rng, case):
    b = int(rng.choice([128, 128, 100, 65, 64, 40, 23, 20, 1]))
    Lb = L(b)
    n = int(rng.choice([0, 1, 63, 64, 65, 4099, 61_706, 300_001]))
    V = int(rng.choice([0, 1, 2, 9, 64, 65, 100, 150]))
    if n * max(V, 1) > 40_000_000:
        V = 9
    in_limbs = Lb if (Lb == 2 and rng.random() < 0.4) else 1
    eng = E.Engine(KEY, b, device=0)

    def vec(limbs):
        a = np.zeros((n, limbs), dtype=np.uint64)
        a[:, 0] = rng.integers(0, 2 ** 64, n, dtype=np.uint64) if b >= 64 else rng.integers(0, 2 ** b, n, dtype=np.uint64)
        if limbs == 2:
            a[:, 1] = rng.integers(0, 2 ** (b - 64), n, dtype=np.uint64)
        return a
    pool = [vec(Lb) for _ in range(4)]
    ins = [vec(in_limbs) for _ in range(min(V, 5))]
    pick = lambda arr, v: arr[v % len(arr)]                            # noqa: E731
    no_minus = rng.random() < 0.5                                       # (round 6: batches without a minus operand take the 120-entry table)
    adds = [None if rng.random() < 0.2 else pick(pool, v) for v in range(V)]
    mins = [None if (no_minus or rng.random() < 0.3) else pick(pool, v + 1) for v in range(V)]
    d_pool = [eng.upload(x) if n else eng.alloc(16) for x in pool]
    d_ins = [eng.upload(x) if n else eng.alloc(16) for x in ins]
    d_in = [pick(d_ins, v) for v in range(V)]
    d_add = [None if adds[v] is None else pick(d_pool, v) for v in range(V)]
    d_min = [None if mins[v] is None else pick(d_pool, v + 1) for v in range(V)]
    outs = [eng.alloc_vec(max(n, 1)) for _ in range(V)]
    dsum = eng.alloc_vec(max(n, 1))
    eng._check(eng._lib.flashe_memset_dev(eng._h, dsum.ptr, 0x3C, dsum.nbytes))
    # round 6: half of the cases through flashe_combine_batch_sum_decrypt_dev (the same pass also decrypts the sum with precomputed masks)
    fused_dec = rng.random() < 0.5
    ddec = eng.alloc_vec(max(n, 1))
    dec_min = rng.random() < 0.5
    if fused_dec:
        eng._check(eng._lib.flashe_memset_dev(eng._h, ddec.ptr, 0x5B, ddec.nbytes))
        eng.combine_batch_sum_decrypt_dev(n, d_in, in_limbs, d_add, None if no_minus and rng.random() < 0.5 else d_min, outs, dsum,
                                          d_pool[2], d_pool[3] if dec_min else None, ddec)
    else:
        eng.combine_batch_sum_dev(n, d_in, in_limbs, d_add, d_min, outs, dsum)
    if n:
        want = [orc.combine(b, pick(ins, v), adds[v], mins[v]) for v in range(V)]
        for v in ([0, V // 2, V - 1] if V else []):
            assert np.array_equal(outs[v].download(np.uint64, n * Lb).reshape(n, Lb), want[v]), ("combsum/out", case, b, n, V, v)
        wsum = orc.aggregate_elem(want, b) if V else np.zeros((n, Lb), dtype=np.uint64)
        assert np.array_equal(dsum.download(np.uint64, n * Lb).reshape(n, Lb), wsum), ("combsum/sum", case, b, n, V, in_limbs)
        if fused_dec:
            wdec = orc.combine(b, wsum, pool[2], pool[3] if dec_min else None)
            assert np.array_equal(ddec.download(np.uint64, n * Lb).reshape(n, Lb), wdec), ("combsum/dec", case, b, n, V, in_limbs, no_minus)
    return f"b={b} n={n} V={V} in_limbs={in_limbs} no_minus={no_minus} fused_dec={fused_dec}"


def _topk_ref(x, res, k):
    """Client.sparsify's selection for one layer (jzf_aggregator.py:585-613): the k largest |x| -- np.argsort of the flattened |x| is
    stable, the LAST k indices are taken, so ties at the threshold go to the higher index -- then sorted locations, values x + residual,
    residual reset at the selected positions."""
    order = np.argsort(np.abs(x), kind="stable")
    sel = np.sort(order[len(x) - k:]) if k else np.zeros(0, dtype=np.int64)
    full = x + res
    new_res = full.copy()
    new_res[sel] = 0
    return sel.astype(np.uint32), full[sel], new_res


def fuzz_sparsify(rng, case):
    dt = np.float32 if rng.random() < 0.7 else np.float64
    single = rng.random() < 0.3
    if single:
        sizes = [int(rng.choice([1, 5, 1023, 1024, 1025, 70_001, 1_100_003, 2_500_000]))]
    else:
        sizes = [int(s) for s in rng.choice([0, 1, 3, 64, 1000, 1024, 1025, 4096, 9408, 36_864, 150_001, 600_000], int(rng.integers(1, 40)))]
    ks = [int(rng.choice([0, 1, max(1, s // 100), max(1, s // 3), s])) if s else 0 for s in sizes]
    quant = rng.random() < 0.5                                          # many ties at the threshold
    layers = []
    for s in sizes:
        x = (rng.standard_normal(s) * 0.05).astype(dt)
        if quant:
            x = (np.round(x * 40) / 40).astype(dt)
        layers.append(x)
    res = [(rng.standard_normal(s) * 0.01).astype(dt) if rng.random() < 0.7 else np.zeros(s, dtype=dt) for s in sizes]
    eng = E.Engine(KEY, 128, device=0)
    flat = np.concatenate(layers) if sum(sizes) else np.zeros(0, dtype=dt)
    fres = np.concatenate(res) if sum(sizes) else np.zeros(0, dtype=dt)
    dx, dr = eng.upload(flat) if flat.size else eng.alloc(16), eng.upload(fres) if fres.size else eng.alloc(16)
    tk = sum(ks)
    dl, dv = eng.alloc(4 * tk + 16), eng.alloc(flat.itemsize * tk + 16)
    if single:
        eng.sparsify_dev(sizes[0], ks[0], dx, dt == np.float64, dr, dl, dv)
    else:
        eng.sparsify_batch_dev(sizes, ks, dx, dt == np.float64, dr, dl, dv)
    eng.sync()
    loc, val, nres = dl.download(np.uint32, tk), dv.download(dt, tk), dr.download(dt, max(flat.size, 0)) if flat.size else np.zeros(0, dtype=dt)
    at = off = 0
    for li, (s, k) in enumerate(zip(sizes, ks)):
        wl, wv, wr = _topk_ref(layers[li], res[li], k)
        if single and k == 0:                                           # (flashe_sparsify_dev with k = 0 is a no-op: the residual stays as it was)
            wr = res[li]
        assert np.array_equal(loc[at:at + k], wl), ("sparsify/loc", case, li, s, k, dt.__name__, quant)
        assert val[at:at + k].tobytes() == wv.tobytes(), ("sparsify/val", case, li, s, k)
        assert nres[off:off + s].tobytes() == wr.tobytes(), ("sparsify/res", case, li, s, k)
        at += k
        off += s
    return f"{dt.__name__} layers={len(sizes)} total={sum(sizes)} k={tk} single={single} ties={quant}"


def fuzz_idx32(rng, case):
    b = int(rng.choice([128, 64, 20]))
    eng = E.Engine(KEY, b, device=0)
    n = int(rng.integers(1, 3000))
    pt = rng.integers(0, 2 ** min(b, 63), n, dtype=np.uint64)
    top = 0xFFFFFFFF
    try:
        eng.encrypt(3, top, E.SCHEME_DOUBLE, 4, pt)
        raise AssertionError(("idx32: accepted", case, b))
    except E.FlasheError as e:
        assert e.code == -22, e
    assert np.array_equal(eng.encrypt(3, top, E.SCHEME_SINGLE, 4, pt), orc.encrypt(KEY, 3, top, "single", 4, b, pt))
    assert np.array_equal(eng.encrypt(3, top - 1, E.SCHEME_DOUBLE, 4, pt), orc.encrypt(KEY, 3, top - 1, "double", 4, b, pt))
    return f"b={b} n={n}"


FAMILIES = {"fixed": fuzz_fixed, "fixed64": fuzz_fixed64, "encsum": fuzz_encsum, "combsum": fuzz_combsum, "sparsify": fuzz_sparsify, "idx32": fuzz_idx32}


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
    print(f"FUZZ_R5_OK {total} cases (seed {seed}, families {','.join(fams)})")


if __name__ == "__main__":
    main()
