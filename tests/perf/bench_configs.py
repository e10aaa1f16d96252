#!/usr/bin/env python3
"""Secondary measurements on one MI355X for the BASELINE configs that are not the headline bench line
(config 3: LeNet-sized + precompute; config 5: top-k sparse; packed aggregate, codec, bit-packing).
Everything is device-resident and timed with HIP events on the engine's stream; results are bit-checked
against the oracle on a prefix where the oracle is fast enough.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flashe_amd.engine import SCHEME_DOUBLE, SCHEME_SINGLE, Engine, telescope  # noqa: E402
from oracle import flashe_oracle as orc  # noqa: E402

KEY = bytes(range(32))


def timeit(eng, fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = eng.event(), eng.event()
    eng.record(e0)
    for _ in range(reps):
        fn()
    eng.record(e1)
    return eng.elapsed_ms(e0, e1) / reps


def config3(out):
    """n = 61 706 (LeNet-5), C = 100, double mask; online vs precomputed masks; b = 128 and b = 23 (m = 5, n_jobs = 16)."""
    n, C = 61706, 100
    res = {}
    for b in (128, 23):
        eng = Engine(KEY, b)
        L = eng.limbs
        rng = np.random.Generator(np.random.PCG64(3))
        pt = rng.integers(0, 2 ** 16, n, dtype=np.uint64)
        dpt = eng.upload(pt)
        stride = (n * L + 1) // 2 * 2                       # ciphertexts equally spaced: the fused reduce + decrypt walks them by stride
        slab = eng.alloc(8 * stride * C)
        dct = [slab.ptr + 8 * stride * c for c in range(C)]
        dadd, dminus, dagg, dout = eng.alloc_vec(n), eng.alloc_vec(n), eng.alloc_vec(n), eng.alloc_vec(n)
        t_enc = timeit(eng, lambda: eng.encrypt_dev(1, 5, SCHEME_DOUBLE, n, 16, dpt, 1, dct[5]))
        t_prep = timeit(eng, lambda: (eng.mask_dev(1, [5], n, 16, dadd), eng.mask_dev(1, [6], n, 16, dminus)))
        t_enc_pre = timeit(eng, lambda: eng.combine_dev(n, dpt, 1, dadd, dminus, dct[5]))
        for c in range(C):
            eng.encrypt_dev(1, c, SCHEME_DOUBLE, n, 16, dpt, 1, dct[c])
        t_enc_batch = timeit(eng, lambda: eng.encrypt_batch_dev(1, list(range(C)), SCHEME_DOUBLE, n, 16, [dpt] * C, 1, dct))
        t_agg = timeit(eng, lambda: eng.aggregate_elem_dev(dct, n, dagg))
        t_dec = timeit(eng, lambda: eng.decrypt_dev(1, [C], [0], n, 16, dagg, dout))
        t_dec_pre = timeit(eng, lambda: eng.combine_dev(n, dagg, L, dadd, dminus, dout))
        t_aggdec = timeit(eng, lambda: eng.aggregate_decrypt_range_dev(1, [C], [0], n, 16, 0, n, dct, dagg, dout))
        eng.decrypt_dev(1, [C], [0], n, 16, dagg, dout)
        got = dout.download(np.uint64, n * L).reshape(n, L)
        assert np.array_equal(got[:, 0], (pt * np.uint64(C)) & np.uint64((1 << min(b, 64)) - 1 if b < 64 else 2 ** 64 - 1))
        got2 = dout.download(np.uint64, n * L).reshape(n, L)
        eng.aggregate_decrypt_range_dev(1, [C], [0], n, 16, 0, n, dct, dagg, dout)
        assert np.array_equal(dout.download(np.uint64, n * L).reshape(n, L), got2)
        ct5 = slab.download(np.uint64, stride * C)[5 * stride:5 * stride + n * L].reshape(n, L)
        assert np.array_equal(ct5, orc.encrypt(KEY, 1, 5, "double", 16, b, pt))
        # the whole round as ONE graph launch (6 kernels captured once)
        def round_calls():
            eng.encrypt_batch_dev(1, list(range(C)), SCHEME_DOUBLE, n, 16, [dpt] * C, 1, dct)
            eng.aggregate_elem_dev(dct, n, dagg)
            eng.decrypt_dev(1, [C], [0], n, 16, dagg, dout)
        round_calls()
        eng.graph_begin()
        round_calls()
        graph = eng.graph_end()
        # every replay is a NEW round: iter shift r makes the captured calls run with iter + r (kernel arguments are frozen into
        # a graph, the shift is read from device memory), so the replayed rounds never reuse a mask stream
        shift = [0]

        def replay():
            shift[0] += 1
            graph.launch(iter_shift=shift[0])
        t_graph = timeit(eng, replay)
        t_calls = timeit(eng, round_calls)
        graph.launch()
        assert np.array_equal(dout.download(np.uint64, n * L).reshape(n, L), got2)
        t_round = min(t_enc_batch + min(t_agg + t_dec, t_aggdec), t_graph)
        res[f"b{b}"] = {"encrypt_us": t_enc * 1e3, "encrypt_all_clients_batched_us": t_enc_batch * 1e3, "prepare_encrypt_us": t_prep * 1e3, "encrypt_precomputed_us": t_enc_pre * 1e3,
                        "aggregate_C100_us": t_agg * 1e3, "decrypt_us": t_dec * 1e3, "decrypt_precomputed_us": t_dec_pre * 1e3,
                        "aggregate_plus_decrypt_fused_us": t_aggdec * 1e3,
                        "round_as_separate_calls_us": t_calls * 1e3, "round_as_one_graph_launch_us": t_graph * 1e3,
                        "round_ms": t_round, "ciphertexts_per_s": C * n / (t_round * 1e-3)}
    out["config3_lenet_C100"] = res


def config5(out):
    """total = 25 557 032 (ResNet-50), k = 1 % per client, C = 50, b = 128, single mask (the sparse path the reference can run)."""
    total, C, b, J, it = 25_557_032, 50, 128, 16, 0
    k = total // 100
    eng = Engine(KEY, b)
    rng = np.random.Generator(np.random.PCG64(2000))
    locs, dloc, dct, zeros = [], [], [], []
    t0 = time.time()
    for c in range(C):
        loc = np.unique(rng.integers(0, total, int(k * 1.02) + 64, dtype=np.uint64))[:k].astype(np.uint32)   # sorted, distinct
        assert len(loc) == k
        locs.append(loc)
        dloc.append(eng.upload(loc))
    dpt = eng.upload(rng.integers(0, 2 ** 64, k, dtype=np.uint64))
    dct = [eng.alloc_vec(k) for _ in range(C)]
    ddense = [eng.alloc_vec(total) for _ in range(C)]
    dagg, dmask, dout = eng.alloc_vec(total), eng.alloc_vec(total), eng.alloc_vec(total)
    gen_s = time.time() - t0
    t_enc = timeit(eng, lambda: eng.encrypt_dev(it, 7, SCHEME_SINGLE, k, J, dpt, 1, dct[7]), reps=10)
    for c in range(C):
        eng.encrypt_dev(it, c, SCHEME_SINGLE, k, J, dpt, 1, dct[c])
    zero = [12345, 0]
    t_expand = timeit(eng, lambda: eng.expand_to_dense_dev(total, k, dloc[7], dct[7], zero, ddense[7]), reps=5)
    for c in range(C):
        eng.expand_to_dense_dev(total, k, dloc[c], dct[c], zero, ddense[c])
    t_agg = timeit(eng, lambda: eng.aggregate_elem_dev(ddense, total, dagg), reps=5)
    t_mask_general = timeit(eng, lambda: eng.sparse_minus_mask_dev(it, dloc, [k] * C, total, J, dmask), reps=5)
    t_mask = timeit(eng, lambda: eng.sparse_minus_mask_dev(it, dloc, [k] * C, total, J, dmask, sorted_lists=True), reps=5)
    t_dec = timeit(eng, lambda: eng.combine_dev(total, dagg, 2, None, dmask, dout), reps=5)
    # spot check against the oracle on client 7's ciphertext
    ct7 = dct[7].download(np.uint64, 2 * k).reshape(k, 2)
    assert np.array_equal(ct7, orc.encrypt(KEY, it, 7, "single", J, b, dpt.download(np.uint64, k)))
    dvals_zero = [zero] * C
    t_sagg_general = timeit(eng, lambda: eng.sparse_aggregate_dev(total, dloc, [k] * C, dct, dvals_zero, dout), reps=5)
    t_sagg = timeit(eng, lambda: eng.sparse_aggregate_dev(total, dloc, [k] * C, dct, dvals_zero, dout, sorted_lists=True), reps=5)
    eng.sparse_aggregate_dev(total, dloc, [k] * C, dct, dvals_zero, dout, sorted_lists=True)
    eng.aggregate_elem_dev(ddense, total, dagg)
    assert np.array_equal(dout.download(np.uint64, 400000), dagg.download(np.uint64, 400000))     # fused == dense path
    idxs = list(range(C))
    t_enc_batch = timeit(eng, lambda: eng.encrypt_batch_dev(it, idxs, SCHEME_SINGLE, k, J, [dpt] * C, 1, dct), reps=5)
    t_round_dense = C * (t_enc + t_expand) + t_agg + t_mask_general + t_dec
    t_round = t_enc_batch + t_sagg + t_mask + t_dec
    out["config5_sparse_top1pct_C50"] = {
        "k": k, "total": total, "encrypt_compact_us": t_enc * 1e3, "expand_to_dense_ms": t_expand, "aggregate_dense_C50_ms": t_agg,
        "aggregate_dense_TBps": 16 * (C + 1) * total / (t_agg * 1e-3) / 1e12, "sparse_minus_mask_ms": t_mask, "decrypt_ms": t_dec,
        "encrypt_all_clients_batched_ms": t_enc_batch, "sparse_aggregate_fused_ms": t_sagg, "sparse_aggregate_unsorted_lists_ms": t_sagg_general,
        "sparse_minus_mask_unsorted_lists_ms": t_mask_general,
        "round_dense_path_ms": t_round_dense, "round_ms": t_round, "sparse_ciphertexts_per_s": C * k / (t_round * 1e-3),
        "host_index_generation_s": gen_s}


def streaming(out):
    """HBM-bound kernels at config-2 size (n = 1e7, b = 128, C = 10)."""
    n, C, b = 10_000_000, 10, 128
    eng = Engine(KEY, b)
    rng = np.random.Generator(np.random.PCG64(1))
    src = [eng.upload(rng.integers(0, 2 ** 64, 2 * n, dtype=np.uint64)) for _ in range(C)]
    dsum, dpk, dun = eng.alloc_vec(n), eng.alloc_vec(n), eng.alloc_vec(n)
    res = {}
    t = timeit(eng, lambda: eng.aggregate_elem_dev(src, n, dsum))
    res["aggregate_elem"] = {"ms": t, "TBps": 16 * (C + 1) * n / (t * 1e-3) / 1e12}
    t = timeit(eng, lambda: eng.aggregate_packed_dev(src, 2 * n, 128 * n, dsum))
    res["aggregate_packed"] = {"ms": t, "TBps": 16 * (C + 1) * n / (t * 1e-3) / 1e12}
    got = dsum.download(np.uint64, 200000)
    want = orc.aggregate_packed([s.download(np.uint64, 200000) for s in src], 64 * 200000)
    assert np.array_equal(got[:199999], want[:199999])          # low limbs are independent of the cut
    t = timeit(eng, lambda: eng.pack_dev(n, src[0], dpk))
    res["pack_b128"] = {"ms": t, "TBps": 32 * n / (t * 1e-3) / 1e12}
    t = timeit(eng, lambda: eng.unpack_dev(n, dpk, dun))
    res["unpack_b128"] = {"ms": t, "TBps": 32 * n / (t * 1e-3) / 1e12}
    t = timeit(eng, lambda: eng.combine_dev(n, src[0], 2, src[1], src[2], dsum))
    res["combine_precomputed"] = {"ms": t, "TBps": 64 * n / (t * 1e-3) / 1e12}
    # codec: fp32 -> u64 quantise, u128 -> f64 unquantise
    x = eng.upload(rng.standard_normal(n).astype(np.float32))
    u = eng.upload(rng.random(n))
    q = eng.alloc(8 * n)
    t = timeit(eng, lambda: eng.quantize_dev(n, x, False, 8.17121, 32, u, q))
    res["quantize_f32"] = {"ms": t, "TBps": 20 * n / (t * 1e-3) / 1e12}
    f = eng.alloc(8 * n)
    t = timeit(eng, lambda: eng.unquantize_dev(n, src[0], 2, 8.17121, 32, 10, f))
    res["unquantize_u128"] = {"ms": t, "TBps": 24 * n / (t * 1e-3) / 1e12}
    e20 = Engine(KEY, 20)
    v20 = e20.upload(rng.integers(0, 2 ** 20, n, dtype=np.uint64))
    p20 = e20.alloc(8 * ((n * 20 + 63) // 64))
    t = timeit(e20, lambda: e20.pack_dev(n, v20, p20))
    res["pack_b20"] = {"ms": t, "TBps": (8 + 2.5) * n / (t * 1e-3) / 1e12}
    # top-k sparsifier at config-5 size: 25 557 032 float32, k = 1 %
    nt = 25_557_032
    xs = eng.upload(rng.standard_normal(nt).astype(np.float32))
    rs = eng.upload(np.zeros(nt, dtype=np.float32))
    kk = nt // 100
    dl, dv = eng.alloc(4 * kk), eng.alloc(4 * kk)
    t = timeit(eng, lambda: eng.sparsify_dev(nt, kk, xs, False, rs, dl, dv), reps=10)
    res["sparsify_f32_25M_top1pct"] = {"ms": t, "TBps_algorithmic": (4 * nt * 6 + 8 * nt) / (t * 1e-3) / 1e12}
    out["streaming_kernels_n1e7"] = res


def main():
    out = {}
    Engine(KEY, 128).selftest()
    config3(out)
    streaming(out)
    config5(out)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
