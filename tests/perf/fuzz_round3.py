#!/usr/bin/env python3
"""Differential fuzzer for the entry points added in round 3, each compared with the CPU oracle / NumPy on random shapes:
  sum     flashe_encrypt_batch_sum_dev      (ciphertexts + the local partial aggregate; fused form and every fallback shape)
  edges   flashe_sparse_double_masks_dev    (run-edge masks from sorted location lists, dense-position counters)
  draws   flashe_mt19937_random_dev         (np.random.random continued from an arbitrary stream position, state handed back)
  twins   flashe_encrypt / _aggregate_elem / _decrypt on host pointers with random chunk sizes of the copy pipeline
  handles DeviceVector blocks recycled through the caching allocator between calls of different sizes
  layers  flashe_sparsify_batch (top-k of every layer of a random model in one set of launches)
  compact flashe_encrypt_batch_u32_dev / flashe_aggregate_decrypt_u32_dev (int_bits <= 32 on uint32 arrays)
  fused   flashe_aggregate_decrypt_range_dev at b <= 64 (the one-launch reduce + decrypt and its fallbacks: sub-ranges, up to 65 operands)
usage: fuzz_round3.py [cases per family] [seed] [families, comma separated]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd import engine as E  # noqa: E402
import flashe_amd.cipher as cm  # noqa: E402
from flashe_amd.cipher import FlasheCipher  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes((7 * i + 3) & 255 for i in range(32))
cm.N_JOBS = 16


def L(b):
    return (b + 63) // 64


def pick_n(rng):
    kind = rng.integers(0, 6)
    if kind == 0:
        return int(rng.integers(1, 300))
    if kind == 1:
        return int(rng.integers(300, 70_000))
    if kind == 2:
        return int(256 * rng.integers(1, 2000) + rng.integers(-2, 3))          # around wave-tile boundaries
    if kind == 3:
        return int(rng.integers(70_000, 1_200_000))
    if kind == 4:
        return int(1024 * 256 * rng.integers(1, 6) + rng.integers(-1, 2))      # around whole-grid boundaries
    return int(rng.integers(1_200_000, 3_000_000))


def fuzz_sum(rng, case):
    b = int(rng.choice([65, 80, 96, 100, 120, 127, 128, 128, 128, 64, 33, 20]))
    C = int(rng.choice([1, 2, 3, 5, 10, 11, 17, 40, 129]))
    n = pick_n(rng)
    if C * n > 24_000_000:
        n = max(1, 24_000_000 // C)
    scheme = "double" if rng.random() < 0.8 else "single"
    i0 = int(rng.integers(0, 1000)) if rng.random() < 0.8 else 2 ** 32 - 2 - C
    idx = [i0 + c for c in range(C)] if rng.random() < 0.85 else [int(v) for v in rng.integers(0, 50, C)]
    it = int(rng.integers(0, 2 ** 32))
    n_jobs = int(rng.choice([1, 3, 8, 16]))
    eng = E.Engine(KEY, b, device=0)
    two = rng.random() < 0.3 and b > 64
    if two:
        pts = [rng.integers(0, 2 ** 64, (n, 2), dtype=np.uint64) for _ in idx]
        for p in pts:
            if b < 128:
                p[:, 1] &= np.uint64((1 << (b - 64)) - 1)
    else:
        pts = [rng.integers(0, 2 ** min(b, 64), n, dtype=np.uint64) for _ in idx]
    dpt = [eng.upload(p) for p in pts]
    dct = [eng.alloc_vec(n) for _ in idx]
    dsum = eng.alloc_vec(n)
    eng.encrypt_batch_sum_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, n_jobs, dpt, 2 if two else 1, dct, dsum)
    want = [orc.encrypt(KEY, it, i, scheme, n_jobs, b, p) for i, p in zip(idx, pts)]
    Lb = L(b)
    for v in sorted(set([0, C - 1] + [int(x) for x in rng.integers(0, C, 3)])):
        assert np.array_equal(dct[v].download(np.uint64, n * Lb).reshape(n, Lb), want[v]), ("sum/ct", case, b, C, n, idx[:3], scheme, v)
    assert np.array_equal(dsum.download(np.uint64, n * Lb).reshape(n, Lb), orc.aggregate_elem(want, b)), ("sum/agg", case, b, C, n, idx[:3], scheme)
    return f"b={b} C={C} n={n} {scheme} two_limb={two}"


def fuzz_edges(rng, case):
    b = int(rng.choice([128, 120, 65, 64, 40, 23, 16, 7]))
    total = int(rng.choice([1, 17, 255, 256, 257, 4096, 4097, 50_000, 300_001]))
    C = int(rng.choice([1, 2, 3, 6, 20, 65]))
    it = int(rng.integers(0, 2 ** 32))
    dens = float(rng.choice([0.0, 0.01, 0.1, 0.5, 1.0]))
    base = np.sort(rng.choice(total, max(1, int(total * dens)), replace=False)) if dens > 0 else np.zeros(0, dtype=np.int64)
    locs = []
    for c in range(C):
        mode = rng.integers(0, 5)
        if mode == 0 or len(base) == 0:
            l = np.sort(rng.choice(total, int(rng.integers(0, min(total, 2000) + 1)), replace=False))
        elif mode == 1:
            l = base
        elif mode == 2:
            l = np.zeros(0, dtype=np.int64)
        else:
            l = np.unique(np.concatenate([base[rng.random(len(base)) < rng.random()], rng.choice(total, max(1, len(base) // 8), replace=False)]))
        locs.append(l.astype(np.uint32))
    ohs = []
    for l in locs:
        a = np.zeros(total, dtype=np.uint8)
        a[l] = 1
        ohs.append(a)
    minus = [ohs[c] & (1 - ohs[c - 1]) if c > 0 else ohs[c] for c in range(C)]
    add = [np.zeros(total, dtype=np.uint8)] + [ohs[c] & (1 - ohs[c + 1]) if c < C - 1 else ohs[c] for c in range(C)]
    want_add, want_minus = orc.sparse_dense_mask(KEY, it, add, total, b), orc.sparse_dense_mask(KEY, it, minus, total, b)
    eng = E.Engine(KEY, b, device=0)
    dloc = [eng.upload(l) if len(l) else eng.alloc(16) for l in locs]
    da, dm = eng.alloc_vec(total), eng.alloc_vec(total)
    eng.sparse_double_masks_dev(it, dloc, [len(l) for l in locs], total, da, dm)
    Lb = L(b)
    assert np.array_equal(da.download(np.uint64, total * Lb).reshape(total, Lb), want_add), ("edges/add", case, b, total, C, dens)
    assert np.array_equal(dm.download(np.uint64, total * Lb).reshape(total, Lb), want_minus), ("edges/minus", case, b, total, C, dens)
    return f"b={b} total={total} C={C} density={dens}"


def fuzz_draws(rng, case):
    seed = int(rng.integers(0, 2 ** 32))
    skip = int(rng.choice([0, 1, 2, 311, 312, 313, 623, 624, 625, int(rng.integers(0, 5000))]))      # doubles drawn first: moves `pos`
    n = int(rng.choice([1, 2, 311, 312, 313, 65535, 65536, 65537, int(rng.integers(1, 400_000)), int(rng.integers(400_000, 6_000_000))]))
    odd = rng.random() < 0.3
    np.random.seed(seed)
    if odd:
        np.random.randint(0, 2 ** 31)                   # one 32-bit draw: the position becomes odd
    if skip:
        np.random.random(skip)
    st = np.random.get_state()
    want = np.random.random(n)
    after = np.random.random(5)
    np.random.set_state(st)
    eng = E.Engine(KEY, 128, device=0)
    du = eng.numpy_random_dev(n)
    got = du.download(np.float64, n)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), ("draws", case, seed, skip, n, odd, int(np.argmax(got != want)))
    assert np.array_equal(np.random.random(5).view(np.uint64), after.view(np.uint64)), ("draws/state", case, seed, skip, n, odd)
    return f"seed={seed} skip={skip} n={n} odd_pos={odd}"


def fuzz_twins(rng, case):
    b = int(rng.choice([128, 128, 100, 64, 20]))
    C = int(rng.choice([1, 2, 4, 10]))
    n = pick_n(rng)
    os.environ["FLASHE_TWIN_CHUNK_MB"] = str(rng.choice([1, 1, 2, 5, 32]))
    os.environ["FLASHE_TWIN_PIPELINE"] = str(rng.choice([0, 1, 1, 1]))
    os.environ["FLASHE_HOST_POOL_PINNED"] = str(rng.choice(["auto", "always", "never"]))
    it = int(rng.integers(0, 2 ** 32))
    ciphers = []
    for c in range(C):
        ci = FlasheCipher(b)
        ci.set_num_clients(C)
        ci.generate_prp_seed(KEY)
        ci.set_iter_index(it)
        ci.idx = c
        ciphers.append(ci)
    pts = [rng.integers(0, 2 ** min(b - 4, 60), n, dtype=np.uint64) for _ in range(C)]
    cts = [ci.encrypt(p) for ci, p in zip(ciphers, pts)]
    want_ct = [orc.encrypt(KEY, it, c, "double", 16, b, p) for c, p in enumerate(pts)]
    Lb = L(b)
    for c in (0, C - 1):
        assert np.array_equal(np.asarray(cts[c]).reshape(n, Lb), want_ct[c]), ("twins/ct", case, b, C, n, c)
    agg = ciphers[0].aggregate(cts)
    assert np.array_equal(np.asarray(agg).reshape(n, Lb), orc.aggregate_elem(want_ct, b)), ("twins/agg", case, b, C, n)
    ciphers[0].set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
    dec = np.asarray(ciphers[0].decrypt(agg)).reshape(n, Lb)
    tot = np.zeros(n, dtype=np.uint64)
    for p in pts:
        tot += p
    m = np.uint64((1 << b) - 1 if b < 64 else 2 ** 64 - 1)
    assert np.array_equal(dec[:, 0] & m, tot & m), ("twins/dec", case, b, C, n)
    return f"b={b} C={C} n={n} chunk={os.environ['FLASHE_TWIN_CHUNK_MB']} MB pipeline={os.environ['FLASHE_TWIN_PIPELINE']} pinned={os.environ['FLASHE_HOST_POOL_PINNED']}"


def fuzz_handles(rng, case):
    b = int(rng.choice([128, 96, 64, 20]))
    C = int(rng.choice([2, 3, 10]))
    it = int(rng.integers(0, 2 ** 32))
    ciphers = []
    for c in range(C):
        ci = FlasheCipher(b)
        ci.set_num_clients(C)
        ci.generate_prp_seed(KEY)
        ci.set_iter_index(it)
        ci.idx = c
        ciphers.append(ci)
    # several rounds of different sizes through the same engines: freed blocks of one round are the next round's results
    for r in range(int(rng.integers(2, 5))):
        n = pick_n(rng) // 2 + 1
        pts = [rng.integers(0, 2 ** min(b - 4, 60), n, dtype=np.uint64) for _ in range(C)]
        hs = [ci.encrypt(p, device=True) for ci, p in zip(ciphers, pts)]
        if rng.random() < 0.5:
            hs[0] = hs[0].to_host()                                  # mixed operands: one host array among the handles
        agg = ciphers[0].aggregate(hs)
        del hs
        ciphers[0].set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
        dec = np.asarray(ciphers[0].decrypt(agg, device=False)).reshape(n, L(b))
        tot = np.zeros(n, dtype=np.uint64)
        for p in pts:
            tot += p
        m = np.uint64((1 << b) - 1 if b < 64 else 2 ** 64 - 1)
        assert np.array_equal(dec[:, 0] & m, tot & m), ("handles", case, b, C, n, r)
    return f"b={b} C={C}"


def fuzz_fused(rng, case):
    """flashe_aggregate_decrypt_range_dev at b <= 64: one launch (small_reduce_decrypt kernels) or the two-launch fallback."""
    b = int(rng.choice([64, 64, 63, 48, 40, 33, 32, 32, 31, 25, 23, 20, 20, 17, 16, 12, 9, 8, 7, 5, 3, 2, 1]))
    n = pick_n(rng) // 3 + 1
    J = int(rng.choice([1, 2, 3, 7, 16, 16, 33, 1000, n + 5]))
    C = int(rng.choice([1, 2, 3, 4, 5, 9, 10, 11, 16, 17, 33, 64, 65]))
    if C * n > 30_000_000:
        n = 30_000_000 // C
    it = int(rng.integers(0, 2 ** 32))
    kind = rng.integers(0, 5)
    add, minus = [([C], [0]), ([int(rng.integers(0, 2 ** 32 - 1))], []), ([int(rng.integers(0, 100))], [int(rng.integers(0, 100))]),
                  ([3, 9], [0, 5]), ([], [1, 2])][kind]
    first = int(rng.integers(0, n)) if rng.random() < 0.5 else 0
    count = int(rng.integers(1, n - first + 1)) if rng.random() < 0.6 else n - first
    keep = rng.random() < 0.5
    eng = E.Engine(KEY, b, device=0)
    cts = [rng.integers(0, 2 ** 64, n, dtype=np.uint64) & np.uint64((1 << b) - 1 if b < 64 else 2 ** 64 - 1) for _ in range(C)]
    if rng.random() < 0.2 and b <= 32:
        for c in cts:
            c |= np.uint64(0xABCD) << np.uint64(40)          # junk above bit 32 of the container: only the value mod 2^b counts
    d = [eng.upload(c) for c in cts]
    out, ao = eng.alloc_vec(count), eng.alloc_vec(count)
    eng.aggregate_decrypt_range_dev(it, add, minus, n, J, first, count, [x.ptr + 8 * first for x in d], ao if keep else None, out)
    m = np.uint64((1 << b) - 1 if b < 64 else 2 ** 64 - 1)
    agg = np.zeros(n, dtype=np.uint64)
    for c in cts:
        agg += c
    agg &= m
    want = orc.combine(b, agg.reshape(-1, 1)[first:first + count], orc.mask_sum(KEY, it, add, n, J, b)[first:first + count],
                       orc.mask_sum(KEY, it, minus, n, J, b)[first:first + count])
    assert np.array_equal(out.download(np.uint64, count).reshape(-1, 1), want), ("fused/out", case, b, n, J, C, add, minus, first, count)
    if keep:
        assert np.array_equal(ao.download(np.uint64, count), agg[first:first + count]), ("fused/agg", case, b, n, J, C, first, count)
    return f"b={b} n={n} J={J} C={C} add={add} minus={minus} first={first} count={count} keep={keep}"


def fuzz_compact(rng, case):
    """The uint32 layout (int_bits <= 32): flashe_encrypt_batch_u32_dev and flashe_aggregate_decrypt_u32_dev against the oracle."""
    b = int(rng.choice([32, 32, 31, 25, 23, 20, 20, 17, 16, 12, 9, 8, 7, 5, 3, 2, 1]))
    n = pick_n(rng) // 3 + 1
    J = int(rng.choice([1, 2, 3, 7, 16, 16, 33, 1000, n + 5]))
    C = int(rng.choice([1, 2, 3, 4, 5, 9, 10, 11, 16, 17, 33, 64]))
    if C * n > 30_000_000:
        n = 30_000_000 // C
    it = int(rng.integers(0, 2 ** 32))
    scheme = "double" if rng.random() < 0.8 else "single"
    i0 = int(rng.integers(0, 1000))
    idx = [i0 + c for c in range(C)] if rng.random() < 0.8 else [int(v) for v in rng.integers(0, 50, C)]
    eng = E.Engine(KEY, b, device=0)
    pts = [rng.integers(0, 2 ** b, n, dtype=np.uint64) for _ in range(C)]
    d32 = [eng.upload(p.astype(np.uint32)) for p in pts]
    c32 = [eng.alloc(4 * n + 16) for _ in range(C)]
    eng.encrypt_batch_u32_dev(it, idx, E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE, n, J, d32, c32)
    want = [orc.encrypt(KEY, it, i, scheme, J, b, p) for i, p in zip(idx, pts)]
    for v in sorted({0, C - 1, int(rng.integers(0, C))}):
        assert np.array_equal(c32[v].download(np.uint32, n), want[v][:, 0].astype(np.uint32)), ("compact/ct", case, b, n, J, C, v, scheme)
    add, minus = [([idx[-1] + 1], [idx[0]]), ([int(rng.integers(0, 2 ** 32 - 1))], []), ([int(rng.integers(0, 100))], [int(rng.integers(0, 100))])][int(rng.integers(0, 3))]
    first = int(rng.integers(0, n)) if rng.random() < 0.5 else 0
    count = int(rng.integers(1, n - first + 1)) if rng.random() < 0.6 else n - first
    ob = int(rng.choice([4, 8]))
    keep = rng.random() < 0.5
    out, ao = eng.alloc(8 * count + 16), eng.alloc(8 * count + 16)
    eng.aggregate_decrypt_u32_dev(it, add, minus, n, J, first, count, [c.ptr + 4 * first for c in c32], ao if keep else None, out, ob)
    agg = orc.aggregate_elem(want, b)
    full = orc.combine(b, agg[first:first + count], orc.mask_sum(KEY, it, add, n, J, b)[first:first + count], orc.mask_sum(KEY, it, minus, n, J, b)[first:first + count])
    dt = np.uint64 if ob == 8 else np.uint32
    assert np.array_equal(out.download(dt, count).astype(np.uint64), full[:, 0]), ("compact/out", case, b, n, J, C, add, minus, first, count, ob)
    if keep:
        assert np.array_equal(ao.download(dt, count).astype(np.uint64), agg[first:first + count, 0]), ("compact/agg", case, b, n, J, C, first, count, ob)
    return f"b={b} n={n} J={J} C={C} {scheme} add={add} minus={minus} first={first} count={count} out_bytes={ob}"


def fuzz_layers(rng, case):
    """flashe_sparsify_batch: the top-k of every layer of a random model in one set of launches against the oracle, layer by layer."""
    dt = np.float32 if rng.random() < 0.7 else np.float64
    L = int(rng.choice([1, 2, 3, 7, 20, 60]))
    sizes = [int(rng.choice([1, 2, 3, 17, 1023, 1024, 1025, 4096, int(rng.integers(1, 3000)), int(rng.integers(3000, 200_000))])) for _ in range(L)]
    if rng.random() < 0.2:
        sizes[int(rng.integers(0, L))] = int(rng.integers(200_000, 3_000_000))
    layers, ks, res = [], [], []
    for n in sizes:
        x = rng.standard_normal(n).astype(dt)
        mode = rng.integers(0, 4)
        if mode == 0 and n > 4:
            x[:: int(rng.integers(2, 9))] = x[0]                                 # ties, often at the threshold
        elif mode == 1:
            x = np.round(x * 4).astype(dt) / 4                                   # few distinct magnitudes, zeros among them
        layers.append(x)
        ks.append(int(rng.choice([1, n, max(1, n // 100), max(1, n // 2), int(rng.integers(1, n + 1))])))
        res.append(rng.standard_normal(n).astype(dt) if rng.random() < 0.8 else np.zeros(n, dtype=dt))
    eng = E.Engine(KEY, 128, device=0)
    got = eng.sparsify_batch(layers, ks, res)
    for i, (x, k, r, (loc, vals, new)) in enumerate(zip(layers, ks, res, got)):
        wl, wv, wr = orc.sparsify(x, k, r)
        assert np.array_equal(loc, wl) and vals.tobytes() == wv.tobytes() and new.tobytes() == wr.tobytes(), ("layers", case, np.dtype(dt).name, i, x.size, k)
    return f"{np.dtype(dt).name} L={L} sizes={sizes[:6]}"


FAMILIES = {"sum": fuzz_sum, "edges": fuzz_edges, "draws": fuzz_draws, "twins": fuzz_twins, "handles": fuzz_handles, "fused": fuzz_fused, "compact": fuzz_compact, "layers": fuzz_layers}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fams = sys.argv[3].split(",") if len(sys.argv) > 3 else list(FAMILIES)
    total = 0
    for f in fams:
        rng = np.random.Generator(np.random.PCG64([seed, sum(map(ord, f))]))
        for case in range(cases):
            what = FAMILIES[f](rng, case)
            total += 1
            if os.environ.get("FUZZ_VERBOSE"):
                print(f, case, what, flush=True)
        print(f"{f}: {cases} cases ok", flush=True)
    print(f"FUZZ_R3_OK {total} cases", flush=True)


if __name__ == "__main__":
    main()
