"""CPU-only: the C-ABI library loads and exports every symbol include/flashe.h declares, its
host-side logic matches the golden vectors, and it refuses to work without a HIP device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_golden

from flashe_amd import _lib, engine

KEY = bytes(range(32))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "flashe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flashe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/flashe.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    assert lib.flashe_abi_version() == 4
    assert [lib.flashe_limbs(b) for b in (0, 1, 64, 65, 128, 129)] == [0, 1, 1, 2, 2, 0]


def test_chunks_match_reference():
    for c in load_golden("mask_streams.json")["cases"]:
        b = engine.chunks(c["n"], c["n_jobs"])
        assert [[b[i], b[i + 1]] for i in range(c["n_jobs"])] == c["chunks"]
    with pytest.raises(engine.FlasheError):
        engine.chunks(5, 0)


def test_telescope_matches_reference_prefixes():
    for c in load_golden("cipher_rounds.json")["cases"]:
        if c["scheme"] != "double":
            continue
        add, minus = engine.telescope(list(c["uploaded"]))
        it = c["iter"]
        assert [(it.to_bytes(4, "big") + i.to_bytes(4, "big")).hex() for i in add] == c["prefix_add"]
        assert [(it.to_bytes(4, "big") + i.to_bytes(4, "big")).hex() for i in minus] == c["prefix_minus"]
    assert engine.telescope([]) == ([], [])
    assert engine.telescope([0, 1, 2, 4]) == ([3, 5], [0, 4])          # SURVEY.md 8c dropout anchor
    assert engine.telescope([0] * 10) == ([1] * 10, [0] * 10)          # notebook cell 14


def test_host_prp_block_matches_anchors():
    g = load_golden("aes_anchors.json")
    f = g["fips197_c3"]
    assert engine.prp_block(bytes.fromhex(f["key"]), bytes.fromhex(f["pt"])).hex() == f["ct"]
    for c in g["blocks"]:
        assert engine.prp_block(KEY, bytes.fromhex(c["block"])).hex() == c["out"]


def test_no_device_means_loud_failure():
    lib = _lib.load()
    cnt = ctypes.c_int(-1)
    lib.flashe_device_count(ctypes.byref(cnt))
    if cnt.value > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(engine.FlasheError) as e:
        engine.Engine(KEY, 128)
    assert e.value.code == -19 and "no CPU fallback" in str(e.value)
    h = ctypes.c_void_p()
    rc = lib.flashe_ctx_create(ctypes.byref(h), (ctypes.c_uint8 * 32)(), 300, 0, None)
    assert rc == -22 and b"int_bits" in lib.flashe_last_error(None)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "flashe_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_product_library_carries_no_probes_or_tuning_knobs():
    """libflashe_hip.so must not be steerable into skipping work: the timing probes (early exits, launches without their AES rounds) and
    the tuning knobs live only in the -DFLASHE_TUNING build (libflashe_hip_tuning.so, loaded by tests/perf/* through FLASHE_LIB_NAME).
    The product binary does not even contain their names."""
    from flashe_amd import _lib
    blob = open(os.path.join(os.path.dirname(_lib.LIB_PATH), "libflashe_hip.so"), "rb").read()
    for name in (b"PROBE", b"FLASHE_CHAIN_TUNE", b"FLASHE_CHAIN_HALF", b"FLASHE_CHAIN_PARTS", b"FLASHE_CHAIN_GRID", b"FLASHE_SMALL_REDUCE_CB",
                 b"FLASHE_SMALL_REDUCE_SPLIT", b"FLASHE_HYBRID_BS_PERMILLE", b"FLASHE_BS16_WAVES", b"FLASHE_SMALL_DIRECT", b"FLASHE_SMALL_LATENCY",
                 b"FLASHE_SMALL_FUSED_REDUCE", b"FLASHE_MT_PARALLEL"):
        assert name not in blob, name
    tuning = os.path.join(os.path.dirname(_lib.LIB_PATH), "libflashe_hip_tuning.so")
    if os.path.exists(tuning):
        assert b"FLASHE_CHAIN_PROBE" in open(tuning, "rb").read()


def test_header_compiles_as_c_and_cxx(tmp_path):
    """include/flashe.h is a plain-C interface: it must compile stand-alone as C11 and as C++17."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "flashe.h")
    for comp, std, ext in (("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "cpp")):
        src = tmp_path / f"t.{ext}"
        src.write_text('#include "flashe.h"\nint main(void) { return FLASHE_OK + (int)sizeof(flashe_ctx *) * 0; }\n')
        subprocess.check_call([comp, std, "-Wall", "-Werror", "-pedantic", "-I", os.path.dirname(hdr), "-c", str(src),
                               "-o", str(tmp_path / f"t_{ext}.o")])


def test_host_result_pool_recycles_only_dead_arrays(monkeypatch):
    """The result arrays of the host-array API come from a recycling pool (engine._HostPool): a block is reused only after EVERY
    view of the array that leased it is gone, two live arrays never share memory, small arrays stay plain (zeroed) NumPy
    allocations, and the parked bytes respect the budget."""
    import gc
    import numpy as np
    from flashe_amd import engine
    pool = engine._HostPool()
    a = pool.empty((300_000, 2), np.uint64)                     # 4.8 MB
    assert a.shape == (300_000, 2) and a.dtype == np.uint64 and a.flags.writeable
    addr = a.ctypes.data
    a[:] = 5
    view = a[100:200, 1]
    del a
    b = pool.empty((300_000, 2), np.uint64)
    assert b.ctypes.data != addr, "a block with a live view was handed out again"
    assert int(view.sum()) == 500
    del view
    gc.collect()
    c = pool.empty((300_000, 2), np.uint64)
    assert c.ctypes.data == addr, "a dead array's block was not recycled"
    small = pool.empty(1000, np.float64)
    assert small.base is None and not small.any()               # an ordinary zeroed array
    # budget: parked bytes never exceed it
    monkeypatch.setenv("FLASHE_HOST_POOL_MB", "6")
    tight = engine._HostPool()
    xs = [tight.empty(500_000, np.uint64) for _ in range(4)]    # 4 MB each
    del xs
    gc.collect()
    assert tight._held <= 6 << 20
    monkeypatch.setenv("FLASHE_HOST_POOL_MB", "0")
    off = engine._HostPool()
    assert off.empty(500_000, np.uint64).base is None


def test_host_result_pool_pins_a_size_class_that_keeps_coming_back(monkeypatch):
    """FLASHE_HOST_POOL_PINNED=auto: the first PIN_AFTER arrays of a size class are pageable, later ones page-locked (here a stand-in
    block type, no GPU), a returning pinned block is preferred over parked pageable ones, `0` never pins, `1` always does, and
    FLASHE_HOST_POOL_WIPE zeroes a block on its way back."""
    import gc
    import numpy as np
    from flashe_amd import engine

    made = []

    class FakePinned(engine._HostPool._Pageable):
        pinned = True

        def __init__(self, cap):
            super().__init__(cap)
            made.append(cap)

    def kinds(pool, k):
        out = []
        for _ in range(k):
            a = pool.empty(400_000, np.uint64)               # 3.2 MB -> one size class
            out.append(a)
        return out

    monkeypatch.setenv("FLASHE_HOST_POOL_PINNED", "auto")
    pool = engine._HostPool()
    monkeypatch.setattr(pool, "_Pinned", FakePinned)
    monkeypatch.setattr(pool, "PIN_AFTER", 3)
    first = kinds(pool, 3)
    assert made == []                                         # leases 1-3: pageable
    del first
    gc.collect()
    assert sum(len(v) for v in pool._free.values()) == 3
    a = pool.empty(400_000, np.uint64)                        # lease 4: the class is pinned from here on, parked pageable blocks stay parked
    assert len(made) == 1
    addr = a.ctypes.data
    del a
    gc.collect()
    b = pool.empty(400_000, np.uint64)                        # the pinned block comes back before any pageable one
    assert b.ctypes.data == addr and len(made) == 1
    for mode, want in (("0", 0), ("1", 2)):
        made.clear()
        monkeypatch.setenv("FLASHE_HOST_POOL_PINNED", mode)
        p2 = engine._HostPool()
        monkeypatch.setattr(p2, "_Pinned", FakePinned)
        keep = kinds(p2, 2)
        assert len(made) == want, mode
        del keep
    monkeypatch.setenv("FLASHE_HOST_POOL_WIPE", "1")
    monkeypatch.setenv("FLASHE_HOST_POOL_PINNED", "0")
    p3 = engine._HostPool()
    x = p3.empty(400_000, np.uint64)
    x[:] = 7
    addr = x.ctypes.data
    del x
    gc.collect()
    y = p3.empty(400_000, np.uint64)
    assert y.ctypes.data == addr and not y.any()


def _build_c_example(tmp_path):
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "c_round")
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_round.c"),
           "-L", os.path.join(ROOT, "flashe_amd"), "-l:libflashe_hip.so", "-Wl,-rpath," + os.path.join(ROOT, "flashe_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_c_example_builds_against_the_header_and_the_library(tmp_path):
    """examples/c_round.c -- a whole round through the C ABI from plain C11, host vectors in and out -- compiles without warnings
    against include/flashe.h and links against the shared library (it RUNS in the GPU suite)."""
    assert os.path.exists(_build_c_example(tmp_path))


def test_mt19937_jump_polynomials_self_check():
    """The jump polynomials behind the parallel np.random.random (x^(S 2^j) mod phi, built on the host from the 135-term characteristic
    polynomial) reproduce the generator: word S + w of a test stream equals the XOR of the words the polynomial selects (no GPU needed)."""
    import time
    from flashe_amd import _lib
    t0 = time.time()
    assert _lib.load().flashe_mt19937_jump_selfcheck() == 0
    assert time.time() - t0 < 20


def test_mt19937_pass_plan_never_exceeds_the_jump_table():
    """a full pass of flashe_mt19937_random_dev from a stream position >= 2 needed 2^12 + 1 substreams and read one
    level past the 12-level jump table (n >= 2^28 - 311).  The pass cap now leaves room for the words in front of the first draw; this
    drives the host-side plan at the sizes where a pass is full, for every kind of starting position."""
    import ctypes
    from flashe_amd import _lib
    lib = _lib.load()
    for n in (1, 65_536, (1 << 28) - 312, (1 << 28) - 311, (1 << 28) - 1, 1 << 28, (1 << 28) + 1, (1 << 29) + 5, 3 * (1 << 28) + 77):
        for pos in (0, 1, 2, 3, 311, 623, 624):
            mx, avail, passes = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
            assert lib.flashe_mt19937_plan(pos, n, ctypes.byref(mx), ctypes.byref(avail), ctypes.byref(passes)) == 0
            assert avail.value == 4096 and 1 <= mx.value <= avail.value, (n, pos, mx.value)
            assert passes.value == -(-n // (((1 << 29) - 624) // 2))
    assert lib.flashe_mt19937_plan(625, 1, ctypes.byref(mx), ctypes.byref(avail), ctypes.byref(passes)) == -22


def test_job_table_marshals_the_job_list_once():
    """Engine.job_table: the flashe_prf_job array of prf_jobs_dev built once (a hundred clients' jobs cost more host time to marshal than
    their launch takes on the device): the same fields prf_jobs_dev would write per call, optional tails included; no device needed."""
    jobs = [(3, 4, 0, 100, None, 0, 0x1000), (7, None, 16, 84, 0x2000, 1, 0x3000), (9, 10, 0, 100, 0x4000, 2, 0x5000, 5, 128, 0x6000)]
    tab = engine.JobTable(engine.Engine, jobs)                  # (Engine._ptr is a static method: the class stands in for an engine)
    assert len(tab) == 3
    a, b, c = tab.arr[0], tab.arr[1], tab.arr[2]
    names = [f[0] for f in engine.PrfJob._fields_]
    get = lambda j, k: getattr(j, names[k])                    # noqa: E731
    # field order of flashe_prf_job (include/flashe.h): add_idx, minus_idx, has_minus, in_limbs, first, count, in, out, n_in, reserved, in_stride, sum_out
    assert [get(a, k) for k in range(6)] == [3, 4, 1, 0, 0, 100] and not get(a, 6) and get(a, 7) == 0x1000
    assert [get(b, k) for k in range(6)] == [7, 0, 0, 1, 16, 84] and get(b, 6) == 0x2000 and get(b, 7) == 0x3000
    assert get(c, 8) == 5 and get(c, 10) == 128 and get(c, 11) == 0x6000


def test_entry_point_index_is_complete_and_current():
    """include/ENTRY_POINTS.md (generated from the header by tools/gen_entry_index.py: section, reference lines cited, tests that call
    it) lists every exported symbol exactly once and is what the generator writes today."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_entry_index.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(os.path.join(ROOT, "include", "ENTRY_POINTS.md")).read()
    listed = re.findall(r"^\| `(flashe_\w+)` \|", text, re.M)
    assert sorted(listed) == sorted(_lib.EXPORTED_SYMBOLS) and len(listed) == len(set(listed))


def test_hot_kernels_keep_their_register_budget():
    """The code objects inside the built library (tools/kernel_resources.py: llvm-readelf on the .hip_fatbin bundles -- the numbers the
    hardware allocates by, and the source of every resource figure in DESIGN.md / profiles/r06_pmc.json): the kernels of the BASELINE
    configurations run four 1,024-thread waves per SIMD (at most 128 VGPRs) WITHOUT scratch, inside the 160 KiB of LDS.  Round 6 hit
    84-132 bytes of scratch per lane twice while restructuring the span kernel (passes 20-70 % slower): this keeps such a build from
    shipping unnoticed.  (The run-time-width instantiations of prf_small_chain_kernel are known to spill 32 bytes: they are the fall-back
    of widths that have no compiled-in kernel.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    res = kernel_resources.resources(os.path.join(ROOT, "flashe_amd", "libflashe_hip.so"))
    assert len(res) > 80
    hot = [k for k in res if any(t in k for t in ("prf_chain_kernel<", "span_prf_kernel<", "reduce_decrypt_ptrs_kernel<", "combine_batch_sum_kernel<",
                                                  "aggregate_elem_kernel<", "small_reduce_decrypt", "span_bounds_kernel<", "span_reduce_kernel<"))]
    hot += [k for k in res if "prf_small_chain_kernel<true, unsigned int, " in k and ", 0>" not in k] + [k for k in res if "prf_small_chain_kernel<true, unsigned long, 64>" in k]
    assert len(hot) >= 30, len(hot)
    for k in hot:
        r = res[k]
        assert r["scratch_bytes_per_lane"] == 0 and r["vgpr_spills"] == 0, (k, r)
        assert r["vgpr"] + r["agpr"] <= 128 and r["lds_bytes_static"] <= 163840, (k, r)
    chain_sum = [r for k, r in res.items() if "prf_chain_kernel<1024, true, false>" in k]
    assert len(chain_sum) == 1 and chain_sum[0]["vgpr"] <= 104 and chain_sum[0]["lds_bytes_static"] == 133632      # the headline launch: 99 VGPRs
