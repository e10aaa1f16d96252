"""BASELINE.json configurations 3, 4 and 5 at FULL size on one MI355X, through the same code paths bench.py --config runs, every
ciphertext (or, where the dense oracle form would not fit the host, an equivalent sparse restatement) compared with the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))
LENET, RESNET50 = 61_706, 25_557_032


@pytest.fixture(scope="module")
def E():
    from flashe_amd import engine
    return engine


def _sum_limbs(vectors, b):
    lo = np.zeros_like(vectors[0])
    hi = np.zeros_like(vectors[0])
    for p in vectors:
        new = lo + p
        hi += (new < lo).astype(np.uint64)
        lo = new
    if b < 64:
        lo &= np.uint64((1 << b) - 1)
    if b <= 64:
        hi[:] = 0
    elif b < 128:
        hi &= np.uint64((1 << (b - 64)) - 1)
    return lo, hi


@pytest.mark.parametrize("b,n_jobs", [(128, 16), (23, 16)])
def test_config3_lenet_100_clients_with_mask_precompute(E, oracle, b, n_jobs):
    """Config 3: n = 61,706 (LeNet-5), C = 100, double mask + mask precompute, at b = 128 and at the reference-style
    int_bits = 16 + ceil(log2 100) = 23 (m = 5, chunk-dependent counters).  Round r: every client runs prepare_encrypt for the
    next iteration (jzf_flashe.py:599-631); round r + 1: encrypt consumes the cached masks (no AES online, :457,483-486), the
    arbiter reduces, prepare_decrypt + decrypt (:633-666, :537-582).  EVERY ciphertext is compared with the oracle's own encrypt;
    the same round through the device-level batch calls bench.py --config 3 uses must give identical ciphertexts."""
    from flashe_amd import FlasheCipher
    from flashe_amd import cipher as cipher_mod
    cipher_mod.N_JOBS = n_jobs
    n, C, it = LENET, 100, 4
    L = 2 if b > 64 else 1
    rng = np.random.Generator(np.random.PCG64(3))
    pts = [rng.integers(0, 2 ** (16 if b < 64 else 64), n, dtype=np.uint64) for _ in range(C)]
    clients = []
    for c in range(C):
        ci = FlasheCipher(b)
        ci.set_num_clients(C)
        ci.generate_prp_seed(KEY)
        ci.set_iter_index(it - 1)
        ci.idx = c
        ci.set_num_params(n)
        ci.prepare_encrypt()                      # masks for iteration `it`
        clients.append(ci)
    cts = []
    for c, ci in enumerate(clients):
        ci.set_iter_index(it)
        assert "add" in ci.next_iter_encrypt_prepared
        ct = ci.encrypt(pts[c] if L == 1 else pts[c])
        assert not ci.next_iter_encrypt_prepared            # consumed
        want = oracle.encrypt(KEY, it, c, "double", n_jobs, b, pts[c])
        got = ct.reshape(n, -1) if ct.dtype == np.uint64 else None
        assert got is not None and np.array_equal(got, want), (b, c)
        cts.append(got)
    agg = clients[0].aggregate([c if L == 2 else c[:, 0] for c in cts])
    agg = agg.reshape(n, -1)
    assert np.array_equal(agg, oracle.aggregate_elem(cts, b))
    ci = clients[7]
    ci.prepare_decrypt()
    ci.set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
    assert ci.index_prefix_for_add == [] and ci.index_prefix_for_minus == []     # all covered by the precomputed masks
    dec = ci.decrypt(agg if L == 2 else agg[:, 0]).reshape(n, -1)
    lo, hi = _sum_limbs(pts, b)
    assert np.array_equal(dec[:, 0], lo) and (L == 1 or np.array_equal(dec[:, 1], hi))
    # device-level form (bench.py --config 3): one chained launch for every mask difference, one batched combine
    eng = E.Engine(KEY, b, device=0)
    dpt = [eng.upload(p) for p in pts]
    masks = [eng.alloc_vec(n) for _ in range(C)]
    dct = [eng.alloc_vec(n) for _ in range(C)]
    dmask, dagg, ddec = eng.alloc_vec(n), eng.alloc_vec(n), eng.alloc_vec(n)
    eng.prf_jobs_dev(it, n, n_jobs, [(c, c + 1, 0, n, None, 0, masks[c]) for c in range(C)] + [(C, 0, 0, n, None, 0, dmask)])
    eng.combine_batch_dev(n, dpt, 1, masks, None, dct)
    eng.aggregate_elem_dev(dct, n, dagg)
    eng.combine_dev(n, dagg, L, dmask, None, ddec)
    for c in range(C):
        assert np.array_equal(dct[c].download(np.uint64, n * L).reshape(n, L), cts[c]), (b, c, "batched")
    assert np.array_equal(ddec.download(np.uint64, n * L).reshape(n, L), dec)
    # round 5, what bench.py --config 3 runs by default: the online encrypts AND the arbiter's reduce of them in one pass
    # (flashe_combine_batch_sum_dev; 100 vectors = two launches that carry the running sum)
    dct2 = [eng.alloc_vec(n) for _ in range(C)]
    dagg2 = eng.alloc_vec(n)
    eng._check(eng._lib.flashe_memset_dev(eng._h, dagg2.ptr, 0x5A, dagg2.nbytes))
    eng.combine_batch_sum_dev(n, dpt, 1, masks, None, dct2, dagg2)
    for c in range(C):
        assert np.array_equal(dct2[c].download(np.uint64, n * L).reshape(n, L), cts[c]), (b, c, "batched + sum")
    assert np.array_equal(dagg2.download(np.uint64, n * L).reshape(n, L), agg), (b, "sum of the batch")
    # round 6, the default of bench.py --config 3: the same pass also decrypts the sum it has just completed with the decrypting party's
    # precomputed mask (flashe_combine_batch_sum_decrypt_dev) -- a hundred clients without a minus operand are ONE launch
    dct3 = [eng.alloc_vec(n) for _ in range(C)]
    dagg3, ddec3 = eng.alloc_vec(n), eng.alloc_vec(n)
    for buf, pat in [(dagg3, 0x5A), (ddec3, 0xC3)] + [(c_, 0x99) for c_ in dct3]:
        eng.memset_dev(buf, pat, buf.nbytes)
    eng.combine_batch_sum_decrypt_dev(n, dpt, 1, masks, None, dct3, dagg3, dmask, None, ddec3)
    for c in range(C):
        assert np.array_equal(dct3[c].download(np.uint64, n * L).reshape(n, L), cts[c]), (b, c, "batched + sum + decrypt")
    assert np.array_equal(dagg3.download(np.uint64, n * L).reshape(n, L), agg), (b, "sum of the batch (fused decrypt)")
    assert np.array_equal(ddec3.download(np.uint64, n * L).reshape(n, L), dec), (b, "decrypt of the sum from the same pass")


def test_config4_resnet50_ten_clients_one_gpu(E, oracle):
    """Config 4 at N = 1: n = 25,557,032 (ResNet-50), C = 10, b = 128, all ten clients on one GPU through ShardedRound (what
    bench.py --config 4 runs; with more GPUs the clients are dealt 2,2,1,1,1,1,1,1 -- tests/test_dist_gloo.py covers that
    exchange).  Round trip == plaintext sum; two clients' FULL ciphertexts and the full aggregate against the oracle."""
    from flashe_amd.dist import HipOps, ShardedRound, deal_clients
    n, C, b, it = RESNET50, 10, 128, 2
    assert [len(x) for x in deal_clients(C, 8)] == [2, 2, 1, 1, 1, 1, 1, 1]
    eng = E.Engine(KEY, b, device=0)
    ops = HipOps(eng)
    pts = [np.random.Generator(np.random.PCG64(3000 + c)).integers(0, 2 ** 64, n, dtype=np.uint64) for c in range(C)]
    refs = [(ops.upload(p), 0) for p in pts]
    rnd = ShardedRound(ops, n, b, deal_clients(C, 1)[0], 16, total_clients=C)
    out = rnd.run(it, refs, 1)
    got = ops.read((out, 0), 2 * n).reshape(n, 2)
    lo, hi = _sum_limbs(pts, b)
    assert np.array_equal(got[:, 0], lo) and np.array_equal(got[:, 1], hi)
    want_agg = None
    for c in (0, 9):
        ct = ops.read(rnd.ct[c], 2 * n).reshape(n, 2)
        assert np.array_equal(ct, oracle.encrypt(KEY, it, c, "double", 16, b, pts[c])), c
    cts = [ops.read(rnd.ct[c], 2 * n).reshape(n, 2) for c in range(C)]
    want_agg = oracle.aggregate_elem(cts, b)
    del cts
    assert np.array_equal(ops.read((rnd.partial, 0), 2 * n).reshape(n, 2), want_agg)
    # the schedule bench.py --config 4 TIMES (partial-agg: the encrypt launch also writes the sum of its ten ciphertexts, the second
    # launch decrypts that one vector): every buffer poisoned first, ALL ten full ciphertexts, the partial aggregate and the result
    for buf in (rnd.ct_all, rnd.partial, out):
        eng.memset_dev(buf, 0xb6, buf.nbytes)
    out_p = rnd.run(it, refs, 1, partial_agg=True)
    for c in range(C):
        want_c = oracle.encrypt(KEY, it, c, "double", 16, b, pts[c])
        assert np.array_equal(ops.read(rnd.ct[c], 2 * n).reshape(n, 2), want_c), (c, "partial-agg schedule")
        del want_c
    assert np.array_equal(ops.read((rnd.partial, 0), 2 * n).reshape(n, 2), want_agg), "partial aggregate written by the encrypt launch"
    assert np.array_equal(ops.read((out_p, 0), 2 * n).reshape(n, 2), got)
    # the fused schedule (chunked chain launches + mask difference) gives the same plaintext aggregate
    side = E.Engine(KEY, b, device=0)
    rnd2 = ShardedRound(HipOps(eng, side), n, b, list(range(C)), 16, total_clients=C)
    out2 = rnd2.run_fused(it, refs, 1, chunks=4)
    assert np.array_equal(rnd2.ops.read((out2, 0), 2 * n).reshape(n, 2), got)


def test_config5_sparse_top1pct_50_clients(E, oracle):
    """Config 5: total = 25,557,032 positions, k = 255,570 (top 1 %) sorted unique locations per client, C = 50, b = 128, the
    single-mask sparse path.  Every client's compact ciphertext vs the oracle; the fused sparse
    aggregate vs a sparse restatement of expand_to_dense + element-wise reduce (jzf_aggregator.py:150-165, :424-430) -- the 50
    dense intermediates (20 GB) are exactly what the kernel avoids; the dense minus-mask vs the oracle; the decrypted vector vs the
    plaintext sums."""
    total, C, b, it, J = RESNET50, 50, 128, 6, 16
    k = total // 100
    eng = E.Engine(KEY, b, device=0)
    rngs = [np.random.Generator(np.random.PCG64(2000 + c)) for c in range(C)]
    locs = [np.sort(r.choice(total, k, replace=False)).astype(np.uint32) for r in rngs]
    vals = [r.integers(0, 2 ** 64, k, dtype=np.uint64) for r in rngs]
    zero = (1 << 31) + 5
    d_loc = [eng.upload(l) for l in locs]
    d_val = [eng.upload(v) for v in vals]
    d_ct = [eng.alloc_vec(k) for _ in range(C)]
    d_agg, d_mask, d_dec = eng.alloc_vec(total), eng.alloc_vec(total), eng.alloc_vec(total)
    eng.encrypt_batch_dev(it, list(range(C)), E.SCHEME_SINGLE, k, J, d_val, 1, d_ct)
    cts = []
    for c in range(C):
        ct = d_ct[c].download(np.uint64, 2 * k).reshape(k, 2)
        assert np.array_equal(ct, oracle.encrypt(KEY, it, c, "single", J, b, vals[c])), c
        cts.append(ct)
    eng.sparse_aggregate_dev(total, d_loc, [k] * C, d_ct, [zero] * C, d_agg, sorted_lists=True)
    # sum_c expand_to_dense(c) restated sparsely: C * zero everywhere, (ct - zero) added at every client's locations
    lo = np.full(total, np.uint64((C * zero) & (2 ** 64 - 1)), dtype=np.uint64)
    hi = np.full(total, np.uint64((C * zero) >> 64), dtype=np.uint64)
    for c in range(C):
        d_lo = cts[c][:, 0] - np.uint64(zero)
        d_hi = cts[c][:, 1] - (cts[c][:, 0] < np.uint64(zero)).astype(np.uint64)
        new = lo[locs[c]] + d_lo
        hi[locs[c]] += d_hi + (new < d_lo).astype(np.uint64)
        lo[locs[c]] = new
    agg = d_agg.download(np.uint64, 2 * total).reshape(total, 2)
    assert np.array_equal(agg[:, 0], lo) and np.array_equal(agg[:, 1], hi)
    # a handful of positions through the oracle's own expand_to_dense (the literal form), on a cut of the vector
    cut = 200_000
    dense = []
    for c in range(C):
        sel = locs[c] < cut
        dense.append(oracle.expand_to_dense(cut, locs[c][sel], cts[c][sel], np.array([zero, 0], dtype=np.uint64), b))
    assert np.array_equal(agg[:cut], oracle.aggregate_elem(dense, b))
    del dense
    eng.sparse_minus_mask_dev(it, d_loc, [k] * C, total, J, d_mask, sorted_lists=True)
    mask = d_mask.download(np.uint64, 2 * total).reshape(total, 2)
    assert np.array_equal(mask, oracle.sparse_minus_mask(KEY, it, locs, total, J, b))
    eng.combine_dev(total, d_agg, 2, None, d_mask, d_dec)
    dec = d_dec.download(np.uint64, 2 * total).reshape(total, 2)
    # what bench.py --config 5 runs: mask construction and decrypt in one pass, the dense mask never reaches HBM
    eng.sparse_decrypt_dev(it, d_loc, [k] * C, total, J, d_agg, d_mask, sorted_lists=True)
    assert np.array_equal(d_mask.download(np.uint64, 2 * total).reshape(total, 2), dec)
    want = np.full(total, np.uint64((C * zero) & (2 ** 64 - 1)), dtype=np.uint64)
    whi = np.zeros(total, dtype=np.uint64)
    for c in range(C):
        d = vals[c] - np.uint64(zero)
        borrow = (vals[c] < np.uint64(zero)).astype(np.uint64)
        new = want[locs[c]] + d
        whi[locs[c]] += (new < d).astype(np.uint64) - borrow
        want[locs[c]] = new
    assert np.array_equal(dec[:, 0], want) and np.array_equal(dec[:, 1], whi)


@pytest.mark.parametrize("args,keys", [
    (["--config", "1"], ["cpu_baseline"]),
    (["--config", "2", "--n", "2300017", "--no-python-baseline"], ["value_two_launch", "value_unchained", "e2e_ms_device_handles", "cpu_baseline", "round_hbm_frac"]),
    (["--config", "2", "--n", "2300017", "--no-partial-agg", "--no-cpu-baseline", "--no-e2e"], ["value_partial_agg"]),
    (["--config", "2", "--n", "2300017", "--schedule", "partial-agg", "--no-cpu-baseline", "--no-e2e"], []),
    (["--config", "2", "--n", "500000", "--schedule", "auto", "--no-cpu-baseline", "--no-e2e"], []),
    (["--config", "3", "--clients", "7", "--no-cpu-baseline"], []),
    (["--config", "5", "--n", "400000", "--clients", "5", "--no-cpu-baseline"], []),
    (["--config", "2", "--bits", "20", "--n", "700001", "--no-cpu-baseline", "--no-e2e"], []),
    (["--config", "2", "--bits", "20", "--layout", "u32", "--n", "700001"], []),
    (["--config", "2", "--n", "500000", "--force-dist", "--schedule", "sequential", "--no-cpu-baseline", "--no-e2e"], []),
    (["--config", "2", "--bits", "20", "--n", "500000", "--force-dist", "--collective", "allreduce", "--schedule", "sequential", "--no-cpu-baseline", "--no-e2e"], []),
])
def test_bench_lines_on_one_gpu(args, keys):
    """bench.py on one GPU, every configuration and schedule at reduced size: the in-run parity gates (round trip AND ciphertexts
    against the oracle) pass, exactly one JSON line comes out and it carries the contract's fields plus `roofline` -- and the
    extra figures the default line reports beside `value`."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--settle-rounds", "2"] + args,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_WAIT_POLICY="passive"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    # stdout carries the JSON line and NOTHING else (with a communicator RCCL prints a version banner: it must land on stderr)
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    rl = d["roofline"]
    assert rl["bound"] in ("hbm", "lds", "mfma") and rl["peak"] == 8000.0 and rl["unit"] == "GB/s" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-9
    assert "bit-exact" in d["config"]["parity"]
    # every line names the library file that produced it and the machine the preflight saw (one rank: no peers; RCCL only with --force-dist)
    cfg = d["config"]
    assert cfg["library"] == "libflashe_hip.so" and len(cfg["library_sha256_16"]) == 16 and cfg["devices_visible"] >= 1
    assert (cfg["rccl_version"] is not None) == ("--force-dist" in args)
    for k in keys:
        assert k in d, k
    if args[:4] == ["--config", "2", "--n", "2300017"] and "--schedule" not in args:
        # the default round at int_bits > 64 is the fastest bit-exact one: the encrypt launch writes the partial aggregate
        assert d["config"]["schedule_name"] == ("sequential" if "--no-partial-agg" in args else "partial-agg")
        assert rl["kernel_key"] == ("prf_chain_kernel" if "--no-partial-agg" in args else "prf_chain_kernel_sum")
    if "--schedule" in args:
        want = args[args.index("--schedule") + 1]
        assert d["config"]["schedule_name"] == want or want == "auto"


def test_config4_full_size_eight_way_both_partitions(tmp_path):
    """BASELINE config 4 (25,557,032 elements, 10 clients, b = 128) in its 8-way shape with REAL kernels: eight ranks share this GPU and
    exchange through the file-based comm double.  Both partitions of SURVEY 8e: clients dealt 2, 2, 1, 1, 1, 1, 1, 1 (all-to-all +
    sliced decrypt + all-gather) and elements sharded (every rank the whole client chain on its eighth, no exchange for the aggregate):
    decrypted aggregate == plaintext sum at full size, ciphertext slices == the oracle's."""
    import subprocess
    import sys
    from conftest import ROOT
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", FLASHE_TEST_SHM_DIR=str(tmp_path), OMP_WAIT_POLICY="passive")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_config4_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {so[-1500:]}{se[-3000:]}"
    assert "CONFIG4_8WAY_OK" in outs[0][0]
