#!/usr/bin/env python3
"""Headline benchmark: ciphertexts/sec for one FLASHE round (C client encrypts + one C-way
aggregate + one decrypt) on a 1e7-element vector, 64-bit plaintext / 128-bit modulus, double
mask -- BASELINE.json config 2 -- with every buffer resident in HBM when the clock starts.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W        (or simply: python bench.py --gpus N)

A "step" is one round.  With N > 1 (config 2) every rank plays C clients (weak scaling: N*C ciphertext
vectors per round), partial aggregates are reduce-scattered over RCCL (grouped send/recv all-to-all +
local mod-add), every rank decrypts its slice and an all-gather returns the plaintext aggregate to all
ranks (flashe_amd/dist.py).  Rank 0 prints ONE JSON line.  Nothing here imports PyTorch: kernels,
device memory and the collectives all go through the C ABI of libflashe_hip.so; under torchrun only
its RANK / WORLD_SIZE / MASTER_PORT environment is used.

--config selects the other BASELINE configurations (same JSON shape): 3 = LeNet-sized model, 100 clients,
mask precompute; 4 = ResNet-50-sized vector, 10 clients dealt (2,2,1,...,1) over the GPUs (strong scaling);
5 = top-1% sparse uploads of that vector, 50 clients.

The JSON carries `roofline` for the dominant kernel (timed live with HIP events on the stream it runs
on) and, at N = 1, `cpu_baseline`: the CPU oracle (a port of the reference algorithm,
oracle/flashe_oracle.c) timed on this host's cores on a bounded sample of the same workload.  The
oracle is only the baseline / checker here, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The oracle's OpenMP team (parity gates, CPU baseline) must SLEEP between its parallel regions: spinning workers burn the
# container's CPU quota and the throttling hits the thread that submits GPU work (measured: 100 launches 0.3 ms -> 4.9 ms).
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("NCCL_DEBUG", "WARN")     # RCCL's own warnings: the first multi-GPU run must not fail silently
os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")      # ... and they (and its version banner) stay off stdout, which carries the ONE JSON line
# RCCL shares device memory between the ranks of a node through dmabuf IPC handles; the legacy IPC mode is not supported by the host
# driver of this pool (hipIpcGetMemHandle: invalid argument).  Must be in the environment before the HIP runtime starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
KEY = bytes(range(32))
RESNET50 = 25_557_032
LENET = 61_706


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4, 5], default=2, help="BASELINE.json configuration (default 2 = the headline)")
    ap.add_argument("--n", type=int, default=None, help="vector length (default: the configuration's)")
    ap.add_argument("--clients", type=int, default=None, help="config 2: clients per GPU; configs 3-5: clients in all")
    ap.add_argument("--bits", type=int, default=128)
    ap.add_argument("--n-jobs", type=int, default=16)
    ap.add_argument("--prf-backend", choices=["auto", "table", "bitslice", "hybrid", "bitslice16"], default="auto")
    ap.add_argument("--pipeline-chunks", type=int, default=None,
                    help="> 0: chunks of the pipelined / fused schedules (reduce, exchange, decrypt of chunk q run on a side stream "
                         "under the launch of chunk q + 1); 0: sequential phases only; default: 3, 4 and 8 take part in the calibration")
    ap.add_argument("--schedule", choices=["default", "auto", "fused", "pipelined", "sequential", "partial-agg"], default="default",
                    help="default: the two-launch round for config 2 on one GPU (the form profiles/ documents kernel by kernel), 'auto' for "
                         "config 4 and whenever ranks exchange; auto: whichever schedule (and chunk count, and number of CUs left free for the "
                         "exchange) is fastest in an untimed calibration on this box / node, after a clock ramp, best of two passes in opposite "
                         "orders; fused: per chunk one launch does every local encrypt plus the decrypt mask difference, the reduce (which then "
                         "yields the plaintext aggregate) and the exchange hide under the next chunk's launch; pipelined: last client's "
                         "encrypt chunked, reduce / exchange / decrypt on a side stream; sequential: all local encrypts in one launch, then "
                         "reduce (+ exchange) fused with the decrypt; partial-agg: the sequential round with the encrypt launch also writing the "
                         "local partial aggregate sum_c ct_c (SURVEY.md section 5: each GPU encrypts and locally mod-adds its share), so that "
                         "the second launch decrypts (or exchanges) ONE vector instead of re-reading C ciphertexts.  With several GPUs the "
                         "sequential round is always timed first and kept as the fallback line")
    ap.add_argument("--deadline", type=float, default=float(os.environ.get("FLASHE_BENCH_DEADLINE_S", "420")),
                    help="N > 1: seconds the start-up + sequential round may take before every rank gives up (RCCL has no timeouts)")
    ap.add_argument("--calibration-deadline", type=float, default=float(os.environ.get("FLASHE_BENCH_CALIBRATION_DEADLINE_S", "150")),
                    help="N > 1: seconds the optional overlapped schedules (calibration + their timed region) may take; when it passes, or "
                         "when any rank raises there, rank 0 prints the sequential line (config.schedule_fallback_reason says why)")
    ap.add_argument("--no-position-sharded", action="store_true", help="config 5, N > 1: skip the round sharded by position ranges (timed beside the replicas)")
    ap.add_argument("--sparse-separate", action="store_true", help="config 5: the encrypts and the sparse aggregate as separate launches (the round-2 .. 4 form) as `value`")
    ap.add_argument("--no-fused-online", action="store_true",
                    help="config 3: the online encrypts and the arbiter's reduce of them as two launches (the round-2 .. 4 form) instead of one pass")
    ap.add_argument("--no-span-bounds", action="store_true",
                    help="config 5: let the sparse aggregate and the sparse decrypt each compute the span bounds of the location lists "
                         "(the round-3 form) instead of computing them once per round")
    ap.add_argument("--no-element-sharded", action="store_true",
                    help="N > 1: skip the second partition (elements instead of clients sharded over the GPUs, SURVEY.md 8e (i)) that is timed "
                         "after the main line and reported beside it as value_element_sharded")
    ap.add_argument("--no-partial-agg", action="store_true",
                    help="keep the separate local reduce in the sequential round instead of letting the encrypt launch write the partial aggregate "
                         "(the default for int_bits > 64 and whenever ranks exchange)")
    ap.add_argument("--collective", choices=["all_to_all", "allreduce"], default="all_to_all",
                    help="how the GPUs' partial aggregates meet in the sequential round: all_to_all = reduce-scatter from point-to-point transfers "
                         "+ local mod-add + all-gather (any --bits); allreduce = ncclAllReduce(uint64, sum) + mask (--bits <= 64 only)")
    ap.add_argument("--layout", choices=["u64", "u32"], default="u64",
                    help="u32 (config 2, --bits <= 32): plaintexts and ciphertexts as uint32 arrays -- the compact layout of the *_u32_dev entry "
                         "points; the default is the ABI's one-limb layout (uint64 per element)")
    ap.add_argument("--no-unchained", action="store_true", help="skip the FLASHE_CHAIN=0 reference measurement (config 2, one GPU)")
    ap.add_argument("--cus-free", type=int, default=None,
                    help="PRF launches leave this many CUs free for the RCCL transfer kernels of the overlapped schedules (default: 0, or "
                         "whichever of 0 / 16 / 32 / 48 calibrates fastest when ranks exchange)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with 1 GPU: still create the RCCL communicator and run the N > 1 exchange path (world size 1)")
    ap.add_argument("--settle-rounds", type=int, default=32,
                    help="untimed rounds run after the in-run parity check, before the warmup steps (GPU clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="elements of the workload the CPU baseline round runs on (default: all of config 2; ~0.2-2 s)")
    ap.add_argument("--no-python-baseline", action="store_true", help="skip the structure-faithful Python baseline (~10-20 s)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the one PCIe-inclusive round through the host-pointer API")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (a container with a
    16-CPU quota on a 256-thread host is throttled, not sped up, by 128 threads -- measured on the GPU box)."""
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]              # cgroup v2
        if quota != "max":
            cpus = min(cpus, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())            # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cpus = min(cpus, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return cpus


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (the only place bench.py touches oracle/): reported beside the GPU number, never the thing measured
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(n_jobs, b, C, host_pts, sample):
    """Oracle (port of the reference arithmetic) on the host cores: full rounds on a bounded sample of the SAME plaintext
    vectors the GPU just processed."""
    import numpy as np
    from oracle import flashe_oracle as orc
    orc.build()
    ns = min(sample, len(host_pts[0]))
    pts = [np.ascontiguousarray(p[:ns]) for p in host_pts[:C]]
    cores = min(orc.num_threads(), usable_cpus())
    orc.set_num_threads(cores)
    orc.mask(KEY, 0, 0, 1000, 1, b)          # table init outside the clock
    Lb = 2 if b > 64 else 1
    # result buffers are allocated and touched before the clock starts, like the GPU's resident buffers
    cts = [np.ones((ns, Lb), dtype=np.uint64) for _ in range(C)]
    agg, dec = np.ones((ns, Lb), dtype=np.uint64), np.ones((ns, Lb), dtype=np.uint64)
    rounds = []
    for rep in range(5):                     # host noise (page placement, other tenants) is large: best AND median reported
        t0 = time.perf_counter()
        for c in range(C):
            orc.encrypt(KEY, 0, c, "double", n_jobs, b, pts[c], out=cts[c])
        t1 = time.perf_counter()
        orc.aggregate_elem(cts, b, out=agg)
        t2 = time.perf_counter()
        orc.decrypt(KEY, 0, [C], [0], n_jobs, b, agg, out=dec)
        t3 = time.perf_counter()
        rounds.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
    best = min(rounds)
    med = sorted(r[0] for r in rounds)[len(rounds) // 2]
    want = np.zeros(ns, dtype=np.uint64)
    for p in pts:
        want += p
    if b < 64:
        want &= np.uint64((1 << b) - 1)
    assert np.array_equal(dec[:, 0], want), "cpu baseline round trip failed"
    return {"value": C * ns / best[0], "value_median": C * ns / med, "unit": "ciphertexts/s", "cores": cores, "kind": "port",
            "sample": f"best (value) and median (value_median) of 5 full rounds (C={C} encrypts + aggregate + decrypt, b={b}, double mask) "
                      f"on the first {ns} elements of every client vector; oracle/flashe_oracle.c, "
                      f"{'AVX-512 VAES x16' if orc.vaes_available() else 'AES-NI x8' if orc.aesni_available() else 'table'} AES-256, "
                      f"OpenMP x{cores}",
            "phases_s": {"encrypt_xC": best[1], "aggregate": best[2], "decrypt": best[3]},
            "round_s_all": [r[0] for r in rounds]}


def _py_chunk(args):
    """One Pool task of the structure-faithful baseline: the mask stream of one chunk, one AES call per block, Python ints
    (what _static_prepare_encrypt does, jzf_flashe.py:48-82; the AES call goes to the oracle's C block function via ctypes
    where the reference calls pycryptodome)."""
    from oracle import flashe_oracle as orc
    begin, end, it, idx_add, idx_minus, b = args
    m = 128 // b
    mask = (1 << b) - 1
    add, minus = [], []
    for blk in range((end - begin - 1) // m + 1):
        ctr = (begin + blk).to_bytes(8, "big")
        sa = int.from_bytes(orc.aes256_encrypt_block(KEY, it.to_bytes(4, "big") + idx_add.to_bytes(4, "big") + ctr), "big")
        sm = int.from_bytes(orc.aes256_encrypt_block(KEY, it.to_bytes(4, "big") + idx_minus.to_bytes(4, "big") + ctr), "big")
        for t in range(min(m, end - begin - blk * m)):
            add.append((sa >> (b * t)) & mask)
            minus.append((sm >> (b * t)) & mask)
    return add, minus


def python_structure_baseline_child(b, C, sample_n):
    """Runs python_structure_baseline in a fresh interpreter (it forks a Pool: never from a process that has initialised the GPU)
    with a hard time limit; None when it fails -- a baseline must not be able to take the benchmark down."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--py-baseline-child", f"{b},{C},{sample_n}"],
                           capture_output=True, text=True, timeout=120, env=dict(os.environ, OMP_NUM_THREADS="1"))
        for line in r.stdout.splitlines():
            if line.startswith('{"value"'):
                return json.loads(line)
        print(f"python baseline child failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
    except Exception as exc:                    # timeout, spawn failure
        print(f"python baseline child: {exc!r}", file=sys.stderr)
    return None


def python_structure_baseline(b, C, host_pts, sample_n, budget_s=25.0):
    """The reference's own structure on this host: numpy object arrays of Python ints, multiprocessing.Pool(cpu_count), one AES call
    per block from Python, `(value + add - minus) & mask` as an object-array expression, reduce(lambda x, y: (x + y) % mod).
    It runs ~1e5 elements/s, so it is measured on `sample_n` elements of each vector and reported per element."""
    import functools
    import multiprocessing as mp
    import numpy as np
    from oracle import flashe_oracle as orc
    orc.build()
    cores = usable_cpus()
    ns = min(sample_n, len(host_pts[0]))
    mod = 1 << b
    d, r = divmod(ns, cores)
    bounds = [(i * (d + 1) if i < r else r * (d + 1) + (i - r) * d) for i in range(cores + 1)]
    t_start = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        def stream(idx_add, idx_minus):
            parts = pool.map(_py_chunk, [(bounds[i], bounds[i + 1], 0, idx_add, idx_minus, b) for i in range(cores) if bounds[i + 1] > bounds[i]])
            return (np.array([v for a, _ in parts for v in a], dtype=object), np.array([v for _, m in parts for v in m], dtype=object))
        t0 = time.perf_counter()
        cts, done = [], 0
        for c in range(C):
            add, minus = stream(c, c + 1)
            cts.append((np.array([int(v) for v in host_pts[c][:ns]], dtype=object) + add - minus) & (mod - 1))
            done += 1
            if time.perf_counter() - t_start > budget_s * 0.6:
                break
        t1 = time.perf_counter()
        agg = functools.reduce(lambda x, y: (x + y) % mod, cts)
        t2 = time.perf_counter()
        add, minus = stream(C, 0)
        dec = (agg + add - minus) & (mod - 1)
        t3 = time.perf_counter()
    if done == C:
        want = [sum(int(p[j]) for p in host_pts[:C]) % mod for j in range(min(ns, 50))]
        assert [int(v) for v in dec[:50]] == want, "python baseline round trip failed"
    enc_per_client = (t1 - t0) / done
    round_s = enc_per_client * C + (t2 - t1) * (C / done) + (t3 - t2)
    return {"value": C * ns / round_s, "unit": "ciphertexts/s", "cores": cores, "kind": "port",
            "sample": f"structure-faithful Python restatement (object arrays, Pool({cores}), one AES call per block, reduce with % mod) on the "
                      f"first {ns} elements of every client vector; {done} of {C} client encrypts measured"
                      + ("" if done == C else ", the rest extrapolated per client") + "; per-element cost is size-independent, so the "
                      "figure extrapolates linearly to the full vector",
            "phases_s": {"encrypt_per_client": enc_per_client, "aggregate": t2 - t1, "decrypt": t3 - t2}, "measured_n": ns}


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def plaintext(global_client, count, b):
    # Generator(PCG64(1000 + c)).integers(0, 2**64, n, uint64)
    import numpy as np
    hi = 2 ** 64 if b >= 64 else 2 ** max(b - 8, 1)
    return np.random.Generator(np.random.PCG64(1000 + global_client)).integers(0, hi, count, dtype=np.uint64)


def sum_mod(vectors, n, b):
    """(lo, hi) limbs of the mod-2^b sum of uint64 vectors."""
    import numpy as np
    lo = np.zeros(n, dtype=np.uint64)
    hi = np.zeros(n, dtype=np.uint64)
    for p in vectors:
        new = lo + p
        hi += (new < lo).astype(np.uint64)
        lo = new
    if b < 64:
        lo &= np.uint64((1 << b) - 1)
    if b < 128:
        hi &= np.uint64((1 << max(b - 64, 0)) - 1) if b > 64 else np.uint64(0)
    return lo, hi


class _NoWatchdog:
    """World size 1: nothing to wait for, nothing to fall back to."""
    exit_code = 0

    def arm(self, seconds, phase):
        pass

    def disarm(self):
        pass

    def finish(self):
        return True

    def abort(self, reason):
        pass


class _CStdoutToStderr:
    """While active, file descriptor 1 points at stderr: RCCL prints its version banner with printf at communicator creation when
    NCCL_DEBUG is set (NCCL_DEBUG_FILE does not move it), and stdout must carry the ONE JSON line and nothing else.  The C stdio
    buffer is flushed before descriptor 1 goes back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            self._libc.fflush(None)
        finally:
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


class PreflightError(RuntimeError):
    """A rank found that this machine cannot carry the launch (message starts with "preflight:"); .info = what it had established."""

    def __init__(self, msg, info):
        super().__init__(msg)
        self.info = info


def preflight(world, local_rank, need_rccl, test_double=False, devices_override=None):
    """What a rank checks BEFORE it creates its engine (inside the rank: never in the launcher, never by re-exec): enough visible devices
    for WORLD_SIZE ranks, LOCAL_RANK among them, direct peer access from its device to every other rank's (what RCCL's point-to-point
    transport over xGMI needs), an RCCL that loads -- and which build of the library it is running on (basename + sha256[:16]: the line
    certifies the file that produced it; FLASHE_LIB_NAME can point at another build).  Returns the keys that go into `config`; raises
    PreflightError("preflight: ...") on the first failed check.  test_double / devices_override: TEST SEAMS (tests/bench_shm.py: several
    ranks share device 0 through a file-based comm double; the override injects a device count)."""
    import ctypes
    import hashlib
    from flashe_amd import _lib
    lib = _lib.load()
    with open(_lib.LIB_PATH, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()[:16]
    info = {"library": os.path.basename(_lib.LIB_PATH), "library_sha256_16": digest, "abi_version": int(lib.flashe_abi_version()),
            "devices_visible": None, "peer_access_ok": None, "rccl_version": None}
    n = ctypes.c_int(0)
    devices = n.value if lib.flashe_device_count(ctypes.byref(n)) == 0 else 0
    if devices_override is not None:
        devices = int(devices_override)
    info["devices_visible"] = devices
    one_device_each = not test_double or devices_override is not None
    if one_device_each and devices < world:
        raise PreflightError(f"preflight: {devices} devices visible, {world} ranks requested", info)
    if local_rank >= max(devices, 0) or local_rank < 0:
        raise PreflightError(f"preflight: LOCAL_RANK {local_rank} but {devices} devices visible", info)
    if world > 1 and not test_double:
        blocked = []
        for peer in range(world):
            can = ctypes.c_int(0)
            if lib.flashe_device_peer_access(local_rank, peer, ctypes.byref(can)) != 0 or not can.value:
                blocked.append(peer)
        info["peer_access_ok"] = not blocked
        if blocked:
            raise PreflightError(f"preflight: device {local_rank} has no peer access to device(s) {blocked}", info)
    if need_rccl and not test_double:
        v = ctypes.c_int(0)
        if lib.flashe_rccl_version(ctypes.byref(v)) != 0:
            raise PreflightError("preflight: librccl.so could not be loaded (or has no ncclGetVersion)", info)
        code = v.value
        info["rccl_version"] = f"{code // 10000}.{code // 100 % 100}.{code % 100} ({code})"
    return info


def main(comm_factory=None, device_override=None, devices_override=None):
    """comm_factory(rank, world) / device_override / devices_override: TEST SEAMS, never set by this script -- tests/bench_shm.py runs this
    very flow with several ranks on ONE GPU by passing a file-based double of RcclComm (tests/shm_comm.py) and device 0 for every rank;
    devices_override injects a visible-device count into the preflight."""
    if len(sys.argv) == 3 and sys.argv[1] == "--py-baseline-child":
        b, C, ns = (int(v) for v in sys.argv[2].split(","))
        print(json.dumps(python_structure_baseline(b, C, [plaintext(c, ns, b) for c in range(C)], ns)), flush=True)
        return
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # launched bare: start the per-GPU processes BEFORE anything touches the GPU (flashe_amd.dist.spawn, no torchrun needed);
        # spawn watches all ranks and stops the rest when one fails
        from flashe_amd.dist import spawn
        sys.exit(spawn(args.gpus, [os.path.abspath(sys.argv[0])] + sys.argv[1:]))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if device_override is None else device_override

    out = {"metric": "ciphertexts/sec (enc+agg+dec), 1e7-elem vector; achieved HBM GB/s fraction", "unit": "ciphertexts/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
           "dtype": "u128" if args.bits > 64 else "u64", "data": "synthetic"}

    # N > 1: RCCL has no timeouts, so every rank runs a watchdog (flashe_amd.dist.Watchdog: per-phase deadline + an abort file any
    # rank can raise).  When it fires, rank 0 prints the best line it already holds -- the plain sequential round is timed FIRST
    # and kept as the fallback -- and every rank leaves through os._exit (an exit, never an exec).
    state = {"fallback": None, "preflight": {}}

    def emit(line):
        """THE one JSON line: whatever produced it, `config` carries the preflight's keys (devices, peer access, RCCL, library hash)."""
        line = dict(line)
        line["config"] = dict(line.get("config") or {}, **state["preflight"])
        print(json.dumps(line), flush=True)

    if world > 1:
        from flashe_amd.dist import Watchdog

        def on_fire(reason):
            print(f"rank {rank}: watchdog: {reason}", file=sys.stderr)
            if rank != 0:
                # (the launcher stops every rank as soon as one has left: give rank 0's watchdog -- same flag, same poll interval -- the
                # moment it needs to print the line before this rank's exit starts that)
                time.sleep(1.0)
                return
            line = state["fallback"]
            if line is not None and state.get("reason_key"):
                line[state["reason_key"]] = reason             # (the main line is complete: the phase that failed came after it)
            elif line is not None:
                line["config"]["schedule_fallback_reason"] = reason
            elif "preflight:" in reason:                       # a rank refused the machine before any engine existed: say exactly that
                line = dict(out, value=None, ms_per_step=None, error=reason[reason.index("preflight:"):])
            else:
                line = dict(out, value=None, ms_per_step=None, error=f"no round completed: {reason}")
            emit(line)
        wd = Watchdog(rank, world, on_fire)
        wd.arm(args.deadline, "start-up (engine, RCCL communicator) and the sequential round")
    else:
        wd = _NoWatchdog()

    try:
        from flashe_amd.dist import HipOps, RcclComm
        from flashe_amd.engine import Engine
        b = args.bits
        cfg = args.config
        n = args.n or {1: 10_000, 2: 10_000_000, 3: LENET, 4: RESNET50, 5: RESNET50}[cfg]
        if cfg == 1 and args.bits == 128:
            b = args.bits = 64                      # config 1 is quoted on a 64-bit modulus
            out["dtype"] = "u64"
        state["preflight"] = preflight(world, local_rank, need_rccl=(world > 1 or args.force_dist), test_double=comm_factory is not None,
                                       devices_override=devices_override)
        eng = Engine(KEY, b, device=local_rank)
        eng.selftest()
        backend = {"auto": 0, "table": 1, "bitslice": 2, "hybrid": 3, "bitslice16": 4}[args.prf_backend]
        eng.set_prf_backend(backend)
        two_streams = (args.pipeline_chunks is None or args.pipeline_chunks > 0) and args.schedule != "sequential" and cfg in (2, 4)
        side = Engine(KEY, b, device=local_rank) if two_streams else None
        if comm_factory is not None:
            comm = comm_factory(rank, world)
        else:
            comm = None
            if world > 1 or args.force_dist:
                with _CStdoutToStderr():
                    comm = RcclComm.from_env(eng)
        ops = HipOps(eng, side, comm)

        if cfg == 2 and args.layout == "u32":
            result = bench_compact(args, n, ops, rank, world, out)
        elif cfg in (2, 4):
            result = bench_dense(args, cfg, n, ops, rank, world, out, wd, state)
        elif cfg == 1:
            result = bench_plumbing(args, n, ops, rank, world, out)
        elif cfg == 3:
            result = bench_precompute(args, n, ops, rank, world, out)
        else:
            result = bench_sparse(args, n, ops, rank, world, out)
        if not wd.finish():                   # the watchdog got there first and is printing / exiting
            time.sleep(30)
            os._exit(wd.exit_code)
        if rank == 0:
            emit(result)
    except BaseException as exc:
        if isinstance(exc, PreflightError):
            state["preflight"] = exc.info
        if world == 1:
            if isinstance(exc, PreflightError):                # one rank: the same legible line instead of a traceback, exit code 3
                emit(dict(out, value=None, ms_per_step=None, error=str(exc)))
                sys.exit(3)
            raise
        # a rank that stops must take the others with it (they would wait in their next collective forever): raise the abort flag,
        # every watchdog fires within a poll interval -- rank 0's prints the fallback line if the sequential round is already in
        if isinstance(exc, PreflightError):
            print(f"rank {rank}: {exc}", file=sys.stderr)      # (a refusal, not a crash: no traceback)
        else:
            import traceback
            traceback.print_exc()
        wd.abort(f"{type(exc).__name__}: {exc}")
        time.sleep(30)
        os._exit(wd.exit_code)
    if comm is not None:
        # the line is out; nothing after this point may keep the process (or its GPU) alive
        if world > 1:
            import threading
            threading.Timer(30.0, lambda: os._exit(0)).start()
        ops.barrier()
        comm.close()
        if world > 1:
            sys.stdout.flush()
            os._exit(0)


def timed_region(ops, K, step):
    """EXACTLY K steps between barrier + device sync on both sides; returns the MAX over ranks of the elapsed seconds."""
    ops.barrier()
    ops.sync()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    ops.sync()
    ops.barrier()
    return ops.allreduce(time.perf_counter() - t0, 0)


def settle(ops, step, seconds=0.1):
    """The parity checks before a timed region leave the GPU idle while the host compares vectors, and its clocks drop; a round of
    well under a millisecond does not bring them back within --warmup steps (tests/perf/compact_encrypt_time.py: the first 13 ms of
    launches run 15-20 % slower).  Rounds for `seconds` of wall clock, in groups of eight, before the W warmup steps proper -- only for
    the configurations without collectives (replicas); bench_dense counts its settle rounds instead (--settle-rounds)."""
    t0, it = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            step(it)
            it += 1
        ops.sync()


def two_event_phases(eng, ev):
    """ev[k] = (before, after) the dominant launch of timed step k -- the ONLY event records inside a timed region (a record costs
    ~3 us on the stream, tests/perf/event_cost.py: four per step were 6 % of config 3's round).  -> array [K][2] of milliseconds: the
    launch itself, and the rest of the step (after[k] -> before[k + 1] in stream order; the last step's rest = the mean of the others)."""
    import numpy as np
    dom = [eng.elapsed_ms(e[0], e[1]) for e in ev]
    rest = [eng.elapsed_ms(ev[k][1], ev[k + 1][0]) for k in range(len(ev) - 1)]
    rest.append(sum(rest) / len(rest) if rest else 0.0)
    return np.array(list(zip(dom, rest)))


def traffic_ratio(kernel_key):
    """Measured HBM bytes per algorithmic byte of a kernel, from the committed rocprofv3 PMC passes (profiles/traffic.json)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get("kernels", {}).get(kernel_key, {})
        # the passes THIS kernel's ratio comes from (the file's top-level "source" is whatever tag was summarised last)
        return ent.get("hbm_bytes_per_algorithmic_byte"), ent.get("measured_in") or tj.get("source")
    except Exception:
        return None, None


XGMI_LINK_GBPS = 153.0      # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, every GPU pair its own link)
HBM_STREAM_GBPS = 5900.0    # what the C-way mod-add reduce streams at on one MI355X (profiles/: aggregate_elem_kernel 5.9 TB/s)


def predict_multi_gpu_ms(n, b, per_rank_clients, world, aes_blocks_per_s, partial=True):
    """DESIGN.md section 5's cost model of the dense round on `world` GPUs, evaluated with THIS run's measured AES rate (blocks/s of the
    chained encrypt launch on rank 0) and one xGMI link per GPU pair at XGMI_LINK_GBPS per direction; collectives NOT overlapped (the
    sequential schedule).  per_rank_clients: how many clients each rank plays (client sharding); element sharding plays all of them on
    n / world elements.  A model to read the first multi-GPU measurement against, not a measurement."""
    L = 2 if b > 64 else 1
    m = 1 if L == 2 else 128 // b
    blocks = lambda streams, elems: streams * (-(-int(elems) // m))                  # noqa: E731
    ms = lambda secs: secs * 1e3                                                      # noqa: E731
    cmax, total = max(per_rank_clients), sum(per_rank_clients)
    piece = -(-n // world)                                                            # elements of one exchanged slice
    piece_bytes = piece * 8 * L
    link = piece_bytes / (XGMI_LINK_GBPS * 1e9) if world > 1 else 0.0                 # every peer's piece travels on its own link, concurrently
    enc = blocks(cmax + 1, n) / aes_blocks_per_s                                      # the busiest rank's chain: C_max + 1 streams
    local_reduce = 0.0 if partial else (cmax + 1) * n * 8 * L / (HBM_STREAM_GBPS * 1e9)
    slice_dec = max(blocks(2, piece) / aes_blocks_per_s, (world + 2) * piece_bytes / (HBM_STREAM_GBPS * 1e9))
    client = {"encrypt_busiest_rank": ms(enc), "local_reduce": ms(local_reduce), "all_to_all": ms(link), "reduce_decrypt_of_the_slice": ms(slice_dec),
              "all_gather": ms(link)}
    client["round"] = sum(client.values())
    el_aes = (blocks(total + 1, piece) + blocks(2, piece)) / aes_blocks_per_s
    elem = {"encrypt_plus_decrypt_of_the_slice": ms(el_aes), "all_gather": ms(link)}
    elem["round_no_gather"] = ms(el_aes)
    elem["round"] = ms(el_aes) + ms(link)
    return {"client_sharded_sequential": client, "element_sharded": elem,
            "inputs": {"aes_blocks_per_s_measured_this_run": aes_blocks_per_s, "xgmi_link_GBps_per_direction": XGMI_LINK_GBPS,
                       "hbm_stream_GBps": HBM_STREAM_GBPS, "world": world, "clients_per_rank": list(per_rank_clients), "n": n, "int_bits": b},
            "note": "DESIGN.md section 5's model (collectives not overlapped, one xGMI link per GPU pair), NOT a measurement"}


# ---- configs 2 and 4: dense double-mask round, clients sharded over ranks ------------------------------------------------
def bench_dense(args, cfg, n, ops, rank, world, out, wd, state):
    import numpy as np
    from oracle import flashe_oracle as orc          # parity gate only (before any timed region)
    from flashe_amd.dist import ShardedRound, deal_clients
    eng = ops.engine
    b, K, W, J = args.bits, args.steps, args.warmup, args.n_jobs
    L = 2 if b > 64 else 1
    is_double = bool(getattr(ops.comm, "IS_TEST_DOUBLE", False))
    if cfg == 2:
        cpr = args.clients or 10                          # weak scaling: every GPU plays `cpr` clients
        total = world * cpr
        mine = list(range(rank * cpr, (rank + 1) * cpr))
        scaling = "weak"
    else:
        total = args.clients or 10                        # strong scaling: the 10 clients are dealt over the GPUs
        mine = deal_clients(total, world)[rank]
        scaling = "strong"
    C = len(mine)
    if args.collective == "allreduce" and b > 64:
        raise SystemExit("--collective allreduce needs --bits <= 64 (RCCL has no 128-bit integer sum)")
    rnd = ShardedRound(ops, n, b, mine, J, rank=rank, world=world, total_clients=total, force_collectives=args.force_dist,
                       collective=args.collective)
    host_pts = {c: plaintext(c, n, b) for c in mine}
    pts = [(ops.upload(host_pts[c]), 0) for c in mine]
    Qs = [max(args.pipeline_chunks, 1)] if args.pipeline_chunks is not None else [3, 4, 8]
    Qbox = {"Q": Qs[len(Qs) // 2]}                       # the chunk count run_schedule uses (the calibration varies it)
    enc_ev = [(eng.event(), eng.event()) for _ in range(K)]
    ph_ev = [[eng.event() for _ in range(2)] for _ in range(K)]       # before / after the encrypt launch of every timed round
    # With an exchange the sequential round sends each rank's partial aggregate: the encrypt launch writes it ("each
    # GPU encrypts and locally mod-adds its share"), which removes the separate local reduce (16 (C + 1) B per element of HBM traffic) from
    # the path the first multi-GPU run is guaranteed to report.  One GPU (round 4): the same form is the default for int_bits > 64 -- it is
    # the fastest bit-exact round (the second launch decrypts ONE vector instead of re-reading C) and `value` reports the fastest; the
    # classic two-launch round is measured beside it (`value_two_launch`).  int_bits <= 64 has no one-launch form of the sum.
    partial = args.schedule == "partial-agg" or (args.schedule in ("default", "auto") and not args.no_partial_agg and (world > 1 or b > 64))

    def run_schedule(schedule, it, k=None):
        """One round.  k = index of the timed step (events recorded) or None (warmup / parity run)."""
        Q = Qbox["Q"]
        if schedule == "fused":
            # one bracketed launch per round (chunk k mod Q): event records are not free on a stream
            evs = [enc_ev[k] if (k is not None and q == k % Q) else None for q in range(Q)]
            return rnd.run_fused(it, pts, 1, chunks=Q, launch_events=evs)
        if schedule == "pipelined":
            return rnd.run_pipelined(it, pts, 1, chunks=Q, batch_events=enc_ev[k] if (k is not None and C > 1) else None)
        if k is None:
            return rnd.run(it, pts, 1, partial_agg=partial)
        eng.record(ph_ev[k][0])
        rnd.encrypt_phase(it, pts, 1, partial_agg=partial)   # one launch: every local client's encrypt (one chain of C + 1 streams)
        eng.record(ph_ev[k][1])
        return rnd.reduce_decrypt_phase(it, partial_agg=partial)   # reduce (+ exchange) fused with the decrypt of its result

    lo, hi = sum_mod((host_pts[c] if c in host_pts else plaintext(c, n, b) for c in range(total)), n, b)   # one vector at a time
    orc.build()
    want_ct = {}

    def ciphertext_ok(check_partial):
        """The ciphertexts the timed kernels write are the reference's: EVERY local client's vector at iter 0 against the oracle's encrypt
        when a rank plays at most 10 clients (first and last otherwise; the round trip alone would pass for ANY mask stream, double masks
        telescope), and -- one rank, sequential round -- the local aggregate against the oracle's element-wise reduce of them
        (jzf_aggregator.py:424-430).  Only the first and the last expected vector stay cached (config 4: 409 MB each)."""
        every = list(range(C)) if C <= 10 else sorted({0, C - 1})
        want_sum = np.zeros((n, L), dtype=np.uint64) if (check_partial and len(every) == C and C) else None
        for c in every:
            want = want_ct.get(c)
            if want is None:
                want = orc.encrypt(KEY, 0, mine[c], "double", J, b, host_pts[mine[c]])
                if c in (0, C - 1):
                    want_ct[c] = want
            if not np.array_equal(ops.read(rnd.ct[c], n * L).reshape(n, L), want):
                return False
            if want_sum is not None:
                want_sum = orc.aggregate_elem([want_sum, want], b)
        if want_sum is not None and not np.array_equal(ops.read((rnd.partial, 0), n * L).reshape(n, L), want_sum):
            return False
        return True

    def parity_ok(res, check_partial=False):
        got = ops.read((res, 0), n * L).reshape(n, L)
        good = np.array_equal(got[:, 0], lo) and (L == 1 or np.array_equal(got[:, 1], hi)) and ciphertext_ok(check_partial and world == 1)
        return ops.allreduce(1.0 if good else 0.0, 1) > 0.5          # every rank must agree on the schedule used

    cus = eng.cu_count

    def configure(cand):
        sched, free, q = cand
        Qbox["Q"] = q
        eng.set_cu_limit(cus - free if free else 0)

    def measure(cand):
        """Settle + W warmup + EXACTLY K timed rounds of one configuration -> (elapsed s, launch ms list, phase ms array or None)."""
        schedule = cand[0]
        configure(cand)
        # The parity check leaves the GPU idle while the host compares 1e7 elements, and its clocks drop: run rounds for ~0.1 s so that
        # the timed region does not start on a cold device even when --warmup is small, then the W warmup steps proper.
        for it in range(args.settle_rounds):          # a fixed count: every rank must issue the same collectives
            run_schedule(schedule, it)
            if it % 8 == 7:
                ops.sync()
        for w in range(W):
            run_schedule(schedule, w)
        elapsed = timed_region(ops, K, lambda k: run_schedule(schedule, k, k))
        if schedule == "sequential":
            enc_pairs = [(p[0], p[1]) for p in ph_ev]      # the batched encrypt launch of every timed round
        else:
            enc_pairs = enc_ev if (schedule == "fused" or C > 1) else []
        enc_ms = [eng.elapsed_ms(e0, e1) for e0, e1 in enc_pairs]
        ph = two_event_phases(eng, ph_ev) if schedule == "sequential" else None
        return elapsed, enc_ms, ph

    rccl_world = ops.comm.rccl_world() if (ops.comm is not None and hasattr(ops.comm, "rccl_world")) else None
    ranks_counted = int(round(ops.allreduce(1.0, 2))) if ops.comm is not None else 1      # a SUM over the communicator: every rank adds 1

    def make_line(cand, elapsed, enc_ms, ph, calibration, parity_all_ranks):
        schedule, cus_free, Q = cand
        ms_per_step = elapsed * 1e3 / K
        pt_bytes = 8
        enc_avg_ms = float(np.mean(enc_ms)) if enc_ms else float("nan")
        chained = os.environ.get("FLASHE_CHAIN", "1") != "0"          # consecutive clients share their PRF streams (every bit width)
        if schedule == "fused":
            # per launch: C encrypt links (u64 plaintext in, L-limb ciphertext out) + the mask-difference job (L limbs out) over one
            # chunk of the vector; a chain of C clients is C + 1 AES streams, the mask difference two more
            elems = n / Q
            alg_bytes = elems * (C * (pt_bytes + 8 * L)) + (elems / world) * 8 * L
            blocks = (C + 1 if chained else 2 * C) * elems + 2 * elems / world
            kernel_key = "prf_chain_kernel"
            kernel_name = (f"prf_chain_kernel<1024> (fused AES-256 PRF + 128-bit add/sub: {C} client encrypts sharing {C + 1} streams + "
                           f"decrypt mask difference, 1/{Q} of the vector per launch)")
        else:
            vec = (C - 1) if (schedule == "pipelined" and C > 1) else C
            alg_bytes = vec * n * (pt_bytes + 8 * L) + (n * 8 * L if (partial and schedule == "sequential") else 0)
            blocks = (vec + 1 if chained else 2 * vec) * (n if L == 2 else -(-n // (128 // b)))
            kernel_key = ("prf_chain_kernel_sum" if (partial and schedule == "sequential") else "prf_chain_kernel") if L == 2 else "prf_small_chain_kernel"
            kernel_name = (f"prf_chain_kernel<1024{', SUM' if (partial and schedule == 'sequential') else ''}> (fused AES-256 PRF + 128-bit add/sub = encrypt: {vec} consecutive clients per launch share "
                           f"{vec + 1} PRF streams, ct_c = pt_c + S_c - S_(c+1)"
                           + (", plus the local partial aggregate sum_c ct_c written by the same launch)" if partial and schedule == "sequential" else ")")) if L == 2 else \
                (f"prf_small_chain_kernel (b <= 64: one AES block = {128 // b} elements, a lane owns its block(s) for all {vec + 1} streams of the "
                 f"{vec}-client chain)")
        achieved = alg_bytes / (enc_avg_ms * 1e-3) / 1e9
        ratio, tsrc = traffic_ratio(kernel_key)
        lookups = (196.1 if chained else 196.5) if L == 2 else 208.0    # b <= 64: one-step counter shortcut only
        lds_peak = 32 * cus * 2.4e9                                      # ds_read_b32: 32 lanes per clock per CU (MI355X_MICROARCH.md, LDS)
        frac_lds = lookups * blocks / (enc_avg_ms * 1e-3) / lds_peak
        # the round's algorithmic bytes on this GPU: C encrypts (pt in, ct out) + the C-way reduce (C in, 1 out) + the
        # decrypt of its result (1 in, 1 out); with several GPUs the reduce / decrypt cover a 1/world slice after the exchange
        round_bytes = n * (C * (pt_bytes + 8 * L) + 8 * L * (C + 1)) + (n / world) * (8 * L * (world + 1 if world > 1 else 0) + 16 * L)
        line = dict(out)
        line.update({
            "value": total * n / (elapsed / K), "ms_per_step": ms_per_step, "scaling": scaling,
            "config": {"workload": f"BASELINE config {cfg}: n={n}-element vector, 64-bit plaintext / {b}-bit modulus, "
                                   + (f"{C} clients per GPU" if cfg == 2 else f"{total} clients dealt {[len(x) for x in deal_clients(total, world)]} over the GPUs")
                                   + f", double mask, n_jobs={J}; round = {total} encrypts + {total}-way aggregate + 1 decrypt"
                                   + (f"; {world} GPUs: all-to-all reduce-scatter (grouped ncclSend/ncclRecv) + sliced decrypt + all-gather" if world > 1 else ""),
                       "n": n, "int_bits": b, "clients_total": total, "clients_this_gpu": C, "mask": "double", "prf_backend": args.prf_backend,
                       "schedule": {"fused": f"{Q} chunks; per chunk one launch = all local encrypts + decrypt mask difference; reduce "
                                             "(-> plaintext aggregate) and exchange hidden on a side stream",
                                    "pipelined": f"reduce / exchange / decrypt chunk-pipelined on a side stream ({Q} chunks)",
                                    "sequential": ("two launches: all local encrypts + their local partial aggregate, then the decrypt of it"
                                                   if partial else "two launches: all local encrypts, then reduce fused with decrypt")}[schedule],
                       "schedule_name": "partial-agg" if (partial and schedule == "sequential") else schedule,
                       "schedule_calibration_ms": calibration, "cus_left_free_for_the_exchange": cus_free,
                       "schedule_fallback_reason": None,
                       "schedule_note": None if calibration or args.schedule != "default" else
                       ("config 2 on one GPU: the encrypt launch also writes the local partial aggregate (SURVEY.md section 5), the second launch "
                        "decrypts that one vector -- the fastest bit-exact round measured (value_two_launch = the classic encrypt, then reduce fused "
                        "with decrypt); --schedule auto also tries the fused and pipelined rounds" if partial else
                        "two-launch round (int_bits <= 64 has no one-launch form of the partial aggregate, or --no-partial-agg)"),
                       "collectives": (getattr(ops.comm, "LABEL", None) or "RCCL through libflashe_hip.so (no PyTorch)") if ops.comm else None,
                       "exchange": (("ncclAllReduce(uint64, sum) + mask" if args.collective == "allreduce" else
                                     "grouped ncclSend / ncclRecv all-to-all + local mod-add + ncclAllGather") if ops.comm else None),
                       "rccl_world": rccl_world, "ranks_counted_by_allreduce": ranks_counted, "ranks_parity_ok": bool(parity_all_ranks),
                       "parity": "bit-exact (on every rank, checked in-run before timing: decrypted aggregate == plaintext sum, "
                                 + ("EVERY local client's ciphertext" if C <= 10 else "the first and last local client's ciphertext")
                                 + " == the oracle's encrypt" + (", the local aggregate == the oracle's reduce of them)" if world == 1 and C <= 10 else ")")},
            "roofline": {"kernel": kernel_name, "kernel_key": kernel_key, "bound": "lds" if L == 2 or 128 // b <= 4 else "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "frac_hbm": achieved / HBM_PEAK_GBPS, "frac_lds": frac_lds,
                         "traffic": ratio * alg_bytes if ratio else None,
                         "traffic_source": (f"{tsrc}: HBM bytes per algorithmic byte measured once with rocprofv3 PMC passes on this kernel "
                                            "(FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE) x this run's algorithmic bytes; not re-measured in-run")
                         if ratio else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": enc_avg_ms, "launches_timed": len(enc_ms),
                         "aes_blocks_per_launch": blocks, "aes_blocks_per_s": blocks / (enc_avg_ms * 1e-3),
                         # LDS lookups per AES block: 11 full rounds x 16 + the final round's 16 + 4 (round 2) + the element's one
                         # counter-dependent lookup of round 1 shared by its C + 1 blocks; half-tile tails take 208
                         "lds_lookup_bound": {"lookups_per_block": lookups, "peak_lookups_per_s_at_2.4GHz": lds_peak,
                                              "achieved_lookups_per_s": lookups * blocks / (enc_avg_ms * 1e-3),
                                              "frac_at_2.4GHz": frac_lds},
                         "note": "integer path: this kernel's roof is the AES rate (LDS lookups: `bound`, `frac_lds`), not HBM; `frac` = `frac_hbm` is "
                                 "the HBM fraction, reported as required"},
            "round_hbm_frac": round_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "round_algorithmic_bytes_this_gpu": round_bytes,
            "phases_ms": ({"round": ms_per_step, "note": "phases overlap in this schedule"} if schedule != "sequential" else
                          {"encrypt_xC": float(ph[:, 0].mean()), "reduce_plus_decrypt": float(ph[:, 1].mean())}),
        })
        return line

    # ---- phase A: the plain two-launch round (collectives on the main stream, nothing overlapped), FIRST -------------------------
    # parity gate before any timing counts; then its K timed rounds.  With several ranks this line is kept as the fallback: whatever
    # happens in the optional schedules afterwards (exception, disagreement, a hang until the deadline), rank 0 still prints it.
    seq = ("sequential", 0, Qbox["Q"])
    configure(seq)
    if not parity_ok(run_schedule("sequential", 0), check_partial=True):
        raise SystemExit(f"rank {rank}: PARITY FAILURE: decrypted aggregate != plaintext sum, or ciphertext != oracle")
    want_explicit = args.schedule in ("fused", "pipelined")
    seq_elapsed = seq_line = None
    if not (want_explicit and world == 1):
        seq_elapsed, seq_enc, seq_ph = measure(seq)
        if rank == 0:
            seq_line = make_line(seq, seq_elapsed, seq_enc, seq_ph, None, True)
        state["fallback"] = seq_line
        wd.exit_code = 0                                   # from here on a valid line exists: a fallback is a success

    # ---- phase B: the overlapped schedules, optional, under their own deadline ---------------------------------------------------
    # default: BASELINE config 2 on one GPU runs the two-launch round -- the form whose per-kernel figures profiles/ holds (one chained
    # launch over the whole vector per round; a calibrated run mixes launch shapes in a trace) -- everything else is calibrated;
    # `--schedule auto` calibrates config 2 on one GPU as well (the fused round is 0-6 % faster there, depending on the box)
    calibrate = args.schedule == "auto" or (args.schedule == "default" and (rnd.exchange or cfg != 2))
    if ops.side is None or args.schedule in ("sequential", "partial-agg"):
        overlapped = []
    elif calibrate:
        overlapped = ["fused", "pipelined"]
    elif want_explicit:
        overlapped = [args.schedule]
    else:
        overlapped = []                                    # config 2 on one GPU, default: the two-launch round only
    if b <= 64:                                            # the one-launch job list of the fused round needs b > 64
        overlapped = list(dict.fromkeys("pipelined" if s_ == "fused" else s_ for s_ in overlapped))
    line = seq_line
    if overlapped:
        wd.arm(args.calibration_deadline, "calibration of the overlapped schedules")
        # An RCCL transfer kernel (36.8 KiB of LDS, 248-256 VGPRs per lane) never shares a CU with a PRF workgroup (128 KiB of LDS), and
        # it starts only when ALL its channels find a CU: beside a PRF launch that fills the device it simply waits for the launch to end
        # (tests/perf/rccl_overlap.py: 16 channels need 16 free CUs, the default configuration 32).  So the schedules that hide the
        # exchange under the next chunk's encrypts are also tried with the PRF launches leaving CUs free; results do not depend on it.
        free_options = [args.cus_free] if args.cus_free is not None else [0, 16, 32, 48] if (rnd.exchange and not is_double) else [0]

        def quick_ms(cand, rounds=8):
            """Untimed-region calibration: ms per round of (schedule, CUs left free, chunks), MAX over ranks."""
            configure(cand)
            for it in range(2):
                run_schedule(cand[0], it)
            return timed_region(ops, rounds, lambda k: run_schedule(cand[0], k)) * 1e3 / rounds

        usable = []
        for s_ in overlapped:
            configure((s_, 0, Qs[len(Qs) // 2]))
            if parity_ok(run_schedule(s_, 0)):
                usable.append(s_)
            elif rank == 0:
                print(f"warning: schedule {s_} fails the parity gate; not used", file=sys.stderr)
        cands = []
        for c in usable:
            if c == "fused":
                cands += [(c, f, q) for q in Qs for f in free_options]
            else:
                cands += [(c, f, Qs[len(Qs) // 2]) for f in free_options]
        chosen, calibration = None, None
        if calibrate and cands:
            cands.append(seq)
            # The parity checks left the device idle and its clock low, and the clock needs tens of milliseconds of load to come back
            # (1.87 -> 2.30 GHz): whatever is measured first would lose.  So: a ramp, then every candidate twice,
            # the second pass in the opposite order, best of the two.
            configure(cands[-1])
            for it in range(40):
                run_schedule(cands[-1][0], it)
            table = {}
            for sweep in (cands, cands[::-1]):
                for cand in sweep:
                    table[cand] = min(table.get(cand, float("inf")), quick_ms(cand))
            chosen = min(table, key=table.get)
            calibration = {f"{c}" + (f", {q} chunks" if c != "sequential" else "") + (f", {f} CUs left free" if f else ""): round(ms, 4)
                           for (c, f, q), ms in table.items()}
            if chosen[0] != "sequential":
                configure(chosen)
                if not parity_ok(run_schedule(chosen[0], 0)):   # the chosen chunk count / CU limit, checked like the schedule itself
                    chosen = seq
        elif cands:
            chosen = cands[0]                                    # an explicit --schedule fused / pipelined
        if chosen is not None and chosen[0] != "sequential":
            elapsed, enc_ms, ph = measure(chosen)
            if rank == 0:
                line = make_line(chosen, elapsed, enc_ms, ph, calibration, True)
                if seq_elapsed is not None:
                    line["sequential_ms_per_step"] = seq_elapsed * 1e3 / K
        elif rank == 0 and line is not None:
            line["config"]["schedule_calibration_ms"] = calibration
        configure(seq)
    # ---- phase C: the OTHER partition of the same job -- elements instead of clients sharded over the GPUs ------------
    # `value` stays north_star's client sharding; this is reported beside it.  Optional and under its own deadline: whatever happens
    # here, the line above is what rank 0 prints.
    if (world > 1 or args.force_dist) and not args.no_element_sharded:                # (the same decision on every rank)
        if rank == 0:
            state["fallback"] = line
        state["reason_key"] = "element_sharded_error"
        wd.arm(args.calibration_deadline, "element-sharded round")
        try:
            extra = element_sharded_round(args, n, b, J, ops, rank, world, total, K, W, (lo, hi), orc)
        except Exception as e:                                  # (a raise on one rank leaves the others to the watchdog: the line survives)
            extra = {"element_sharded_error": f"{type(e).__name__}: {e}"}
            wd.abort(f"element-sharded round raised on rank {rank}: {e}")
        if rank == 0 and line is not None:
            line.update(extra)
        state["reason_key"] = None
    if rank == 0 and line is not None and (world > 1 or args.force_dist):
        # what the schedules that ran SHOULD cost by DESIGN.md section 5's model, with this run's own AES rate: lets a reader of the first
        # real multi-GPU line tell "the exchange is slow" from "the partition has a low ceiling" without the design document
        per_rank = [len(x) for x in deal_clients(total, world)] if cfg != 2 else [C] * world
        rate = (seq_line or line)["roofline"]["aes_blocks_per_s"]
        pred = predict_multi_gpu_ms(n, b, per_rank, world, rate, partial=partial)
        line["predicted_ms"] = pred
        seq_ms = line.get("sequential_ms_per_step") or (line["ms_per_step"] if line["config"]["schedule_name"] in ("sequential", "partial-agg") else None)
        ratios = {"schedule_that_ran_over_sequential_model": line["ms_per_step"] / pred["client_sharded_sequential"]["round"]}
        if seq_ms:
            ratios["sequential_over_model"] = seq_ms / pred["client_sharded_sequential"]["round"]
        for key, mk in (("ms_per_step_element_sharded", "round"), ("ms_per_step_element_sharded_no_gather", "round_no_gather")):
            if line.get(key):
                ratios[key.replace("ms_per_step_", "") + "_over_model"] = line[key] / pred["element_sharded"][mk]
        line["scale_check"] = {"rccl_world_is_n": rccl_world == world, "ranks_counted_by_allreduce_is_n": ranks_counted == world,
                               "ranks_parity_ok": bool(line["config"].get("ranks_parity_ok")), "measured_over_predicted": ratios,
                               "how_to_read": "ratios near 1 = the run behaves as modelled; `value` is north_star's client sharding, whose model "
                                              "ceiling at 8 GPUs is ~2.5x for config 4 (stream sharing needs consecutive clients on one GPU); "
                                              "value_element_sharded is the partition that scales"}
    wd.arm(args.deadline, "closing")
    if rank != 0:
        return None
    if line is None:
        raise SystemExit(f"--schedule {args.schedule} did not produce a usable round here (it needs a side stream and must pass the parity gate)")

    if world == 1:
        hp = [host_pts[c] for c in mine]
        if cfg == 2 and not args.no_unchained and args.schedule in ("default", "sequential"):
            line.update(unchained_round(args, n, b, J, mine, total, pts, K))
            # the OTHER form of the same round beside `value` (parity first; measured outside the timed region)
            other = not partial
            res = rnd.run(0, pts, 1, partial_agg=other)
            if not parity_ok(res, check_partial=True):
                raise SystemExit("PARITY FAILURE: " + ("partial-agg" if other else "two-launch") + " round")
            for it in range(12):
                rnd.run(it, pts, 1, partial_agg=other)
            s_o = timed_region(ops, K, lambda k: rnd.run(k, pts, 1, partial_agg=other))
            if other:
                line.update({"ms_per_step_partial_agg": s_o * 1e3 / K, "value_partial_agg": total * n / (s_o / K),
                             "partial_agg_note": "the same round with the encrypt launch also writing the local partial aggregate (sum of its C "
                                                 "ciphertexts, SURVEY.md section 5), so that the second launch decrypts one vector instead of "
                                                 "re-reading C; measured here outside the timed region"})
            else:
                line.update({"ms_per_step_two_launch": s_o * 1e3 / K, "value_two_launch": total * n / (s_o / K),
                             "two_launch_note": "the classic round -- every local encrypt in one chained launch, then the C-way reduce fused with "
                                                "the decrypt of its result (re-reads the C ciphertexts) -- measured here outside the timed region; "
                                                "`value` is the partial-aggregate form (--no-partial-agg makes this one the timed round)"})
        if not args.no_e2e:
            line["e2e_ms_incl_pcie"], line["e2e_first_round_ms"] = e2e_round_ms(eng, hp, n, b, J)
            line["e2e_note"] = ("one round through the host-pointer twins (flashe_encrypt x C, flashe_aggregate_elem, flashe_decrypt): pageable "
                                "caller vectors, results in the engine's recycled host arrays, every call H2D + kernel + D2H (chunk-pipelined when "
                                "the result array is page-locked), calls back to back; the first round also allocates the staging blocks and the "
                                "result arrays, later rounds reuse them; never `value`")
            line["e2e_ms_device_handles"], line["e2e_device_handles_first_round_ms"] = e2e_handles_ms(hp, n, b, J)
            line["e2e_device_handles_note"] = ("the same round through the drop-in class with results kept in HBM between the calls: C x "
                                               "FlasheCipher.encrypt(host plaintext, device=True) -> aggregate(handles) -> decrypt(handle, "
                                               "device=False): C uploads of 8-byte plaintexts, one download; never `value`")
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(J, b, C, hp, args.cpu_sample)
            if not args.no_python_baseline:
                pyb = python_structure_baseline_child(b, C, 1_000_000)  # n = 1e6 per phase; regenerates the plaintext prefix from the seeds
                if pyb:
                    line["cpu_baseline_python"] = pyb
    return line


def element_sharded_round(args, n, b, J, ops, rank, world, total, K, W, want, orc):
    """The same job with ELEMENTS sharded over the ranks (flashe_amd.dist.ShardedRound(shard="elements")): every rank
    runs the full chain of `total` clients (total + 1 PRF streams per element -- the stream sharing that client sharding loses when
    clients are spread thin) on its own slice of the vectors, reduces and decrypts that slice; no exchange for the aggregate, one
    all-gather of the decrypted slices (timed with and without it).  Parity gate as for the main line (result == plaintext sum, first
    and last client's ciphertext slice == the oracle's), then settle + W warmup + EXACTLY K timed rounds, MAX over ranks."""
    import numpy as np
    from flashe_amd.dist import ShardedRound
    L = 2 if b > 64 else 1
    rnd = ShardedRound(ops, n, b, total, J, rank=rank, world=world, force_collectives=args.force_dist, shard="elements")
    first, count = rnd.element_range()
    pts = []
    for c in range(total):
        p = plaintext(c, n, b)
        pts.append((ops.upload(p[first:first + count]) if count else ops.alloc(2), 0))
        if c in (0, total - 1) and count:
            wct = orc.encrypt(KEY, 0, c, "double", J, b, p)[first:first + count]
            if c == 0:
                want_first = wct
            want_last = wct
    res = rnd.run(0, pts, 1, partial_agg=True)
    got = ops.read((res, 0), n * L).reshape(n, L)
    lo, hi = want
    good = np.array_equal(got[:, 0], lo) and (L == 1 or np.array_equal(got[:, 1], hi))
    if count:
        good = good and np.array_equal(ops.read(rnd.ct[0], count * L).reshape(count, L), want_first)
        good = good and np.array_equal(ops.read(rnd.ct[total - 1], count * L).reshape(count, L), want_last)
    if not ops.allreduce(1.0 if good else 0.0, 1) > 0.5:
        return {"element_sharded_error": "parity gate failed"}
    out = {}
    for key, gather in (("", True), ("_no_gather", False)):
        for it in range(min(args.settle_rounds, 16) + W):
            rnd.run_elements(it, pts, 1, partial_agg=True, gather=gather)
        s_el = timed_region(ops, K, lambda k: rnd.run_elements(k, pts, 1, partial_agg=True, gather=gather))
        out["ms_per_step_element_sharded" + key] = s_el * 1e3 / K
        out["value_element_sharded" + key] = total * n / (s_el / K)
    out["element_sharded_note"] = (f"the same {total}-client job with ELEMENTS sharded over the {world} GPUs (SURVEY.md 8e (i)): every GPU runs the whole "
                                   f"client chain ({total + 1} PRF streams) on {rnd.slice} elements of every vector, reduces and decrypts its slice; no "
                                   "exchange for the aggregate, one all-gather of the decrypted slices (`_no_gather`: every GPU keeps its slice); "
                                   "parity-gated like `value` (result == plaintext sum, ciphertext slices == the oracle's); `value` stays north_star's "
                                   "client sharding")
    return out


def unchained_round(args, n, b, J, mine, total, pts, K):
    """The same K rounds with stream sharing OFF (FLASHE_CHAIN=0 on a second ctx: every client computes both of its streams, what a
    GPU that hosts only one client pays), measured outside the timed region and reported beside `value`."""
    from flashe_amd.dist import HipOps, ShardedRound
    from flashe_amd.engine import Engine
    old = os.environ.get("FLASHE_CHAIN")
    os.environ["FLASHE_CHAIN"] = "0"
    try:
        eng_u = Engine(KEY, b, device=0)
    finally:
        if old is None:
            os.environ.pop("FLASHE_CHAIN", None)
        else:
            os.environ["FLASHE_CHAIN"] = old
    ops_u = HipOps(eng_u, None, None)
    rnd_u = ShardedRound(ops_u, n, b, mine, J, rank=0, world=1, total_clients=total)
    for it in range(12):
        rnd_u.run(it, pts, 1)
    s = timed_region(ops_u, K, lambda k: rnd_u.run(k, pts, 1))
    eng_u.close()
    return {"ms_per_step_unchained": s * 1e3 / K, "value_unchained": total * n / (s / K),
            "unchained_note": "the same round with FLASHE_CHAIN=0: every client computes both of its PRF streams (2 C instead of C + 1 AES "
                              "blocks per element-position), i.e. what C GPUs hosting one client each would pay per client"}


def e2e_round_ms(eng, host_pts, n, b, J):
    """The same round with every operand starting and ending in HOST memory (what a caller that keeps its vectors on the
    host pays): PCIe Gen5 transfers included.  Returns (steady, first): the first round also pays for the device staging blocks
    and for the page faults of fresh result arrays; later rounds reuse both (flashe_amd.engine._HostPool, abi.hip Tmp)."""
    import numpy as np
    from flashe_amd.engine import SCHEME_DOUBLE
    C = len(host_pts)
    times = []
    for _ in range(5):          # the result pool page-locks a size class once it keeps coming back (engine._HostPool): rounds 2-3 pay for that
        t0 = time.perf_counter()
        cts = [eng.encrypt(0, c, SCHEME_DOUBLE, J, host_pts[c]) for c in range(C)]
        agg = eng.aggregate_elem(cts)
        dec = eng.decrypt(0, [C], [0], J, agg)
        times.append((time.perf_counter() - t0) * 1e3)
        lo, _ = sum_mod(host_pts, n, b)
        assert np.array_equal(dec[:, 0], lo), "host-pointer round trip failed"
        del cts, agg, dec
    return min(times[3:]), times[0]


def e2e_handles_ms(host_pts, n, b, J):
    """The round through flashe_amd.FlasheCipher with DeviceVector handles between encrypt, aggregate and decrypt (what a caller of the
    drop-in API does to keep ciphertexts off the PCIe bus).  Returns (steady, first) milliseconds."""
    import numpy as np
    from flashe_amd import cipher as cm
    C = len(host_pts)
    old = cm.N_JOBS
    cm.N_JOBS = J
    try:
        clients = []
        for c in range(C):
            ci = cm.FlasheCipher(b)
            ci.set_num_clients(C)
            ci.generate_prp_seed(KEY)
            ci.set_iter_index(0)
            ci.idx = c
            clients.append(ci)
        times = []
        lo, _ = sum_mod(host_pts, n, b)
        for _ in range(4):
            t0 = time.perf_counter()
            handles = [clients[c].encrypt(host_pts[c], device=True) for c in range(C)]
            agg = clients[0].aggregate(handles)
            clients[0].set_idx_list(raw_idx_list=list(range(C)), mode="decrypt")
            dec = clients[0].decrypt(agg, device=False)
            times.append((time.perf_counter() - t0) * 1e3)
            assert np.array_equal(np.asarray(dec).reshape(n, -1)[:, 0], lo), "device-handle round trip failed"
            del handles, agg, dec
        return min(times[1:]), times[0]
    finally:
        cm.N_JOBS = old


# ---- config 1: the reference's own CPU-runnable case (plumbing): fp32 -> 32-bit quantise -> 64-bit modulus, 2 clients, single mask ---
def bench_plumbing(args, n, ops, rank, world, out):
    """Per round: both clients quantise + encrypt their fp32 vector (one fused launch each, stochastic-rounding draws resident), the
    arbiter adds the two ciphertexts, the result is decrypted + unquantised (one fused launch).  1e4 elements: launch bound; the line
    exists so that every BASELINE configuration has one, the parity of this configuration is tests/golden/config1.npz."""
    import numpy as np
    from oracle import flashe_oracle as orc          # parity gate only
    from flashe_amd.engine import SCHEME_SINGLE
    from flashe_amd.quantize import ACIQ
    eng = ops.engine
    b, K, W, J = args.bits, args.steps, args.warmup, args.n_jobs
    C, eb = args.clients or 2, 32
    L = 2 if b > 64 else 1
    alpha = float(ACIQ(eb).get_alpha_gaus_direct(1.0))
    rng = np.random.Generator(np.random.PCG64(7))
    xs = [rng.standard_normal(n).astype(np.float32) for _ in range(C)]
    us = [rng.random(n) for _ in range(C)]
    dx, du = [ops.upload(x) for x in xs], [ops.upload(u) for u in us]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    agg, res = eng.alloc_vec(n), eng.alloc(8 * n)
    ev = [[eng.event() for _ in range(2)] for _ in range(K)]

    def step(it, k=None):
        if k is not None:
            eng.record(ev[k][0])
        for c in range(C):
            eng.quantize_encrypt_dev(it, c, SCHEME_SINGLE, n, J, dx[c], False, alpha, eb, du[c], cts[c])
        if k is not None:
            eng.record(ev[k][1])
        eng.aggregate_elem_dev(cts, n, agg)
        eng.decrypt_unquantize_dev(it, [], list(range(C)), n, J, agg, alpha, eb, C, res)

    step(0)
    orc.build()
    q = [orc.quantize(xs[c], alpha, eb, us[c]) for c in range(C)]
    want_ct = [orc.encrypt(KEY, 0, c, "single", J, b, q[c]) for c in range(C)]
    for c in range(C):
        assert np.array_equal(cts[c].download(np.uint64, n * L).reshape(n, L), want_ct[c]), f"PARITY FAILURE client {c}"
    want = orc.unquantize(orc.decrypt(KEY, 0, [], list(range(C)), J, b, orc.aggregate_elem(want_ct, b)), alpha, eb, C)
    assert res.download(np.float64, n).tobytes() == want.tobytes(), "PARITY FAILURE (decrypt + unquantise)"
    settle(ops, step)
    for it in range(max(W, 3)):
        step(it)
    elapsed = timed_region(ops, K, lambda k: step(k, k))
    if rank != 0:
        return None
    ph = two_event_phases(eng, ev)
    enc_ms = float(ph[:, 0].mean())
    alg_bytes = C * n * (4 + 8 + 8 * L)                   # fp32 value + its draw in, ciphertext out
    out.update({
        "value": world * C * n / (elapsed / K), "ms_per_step": elapsed * 1e3 / K, "scaling": "weak",
        "config": {"workload": f"BASELINE config 1: n={n}-element random fp32 vector, {eb}-bit quantise, {b}-bit modulus, {C} clients, single mask, "
                               f"n_jobs={J}; step = {C} x (quantise + encrypt, one fused launch) + aggregate + (decrypt + unquantise, one fused launch)"
                               + ("; independent replicas per GPU" if world > 1 else ""),
                   "n": n, "int_bits": b, "clients_total": C, "mask": "single",
                   "parity": "bit-exact (ciphertexts and the unquantised result vs the oracle, checked in-run; the reference-generated fixture of "
                             "this configuration is tests/golden/config1.npz)"},
        "roofline": {"kernel": "prf_small_jobs_kernel (AES-256 PRF fused with the quantiser front end: fp32 -> stochastic rounding -> + mask)",
                     "bound": "hbm", "achieved": alg_bytes / (enc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_bytes / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes / C,
                     "avg_launch_ms": enc_ms / C, "launches_timed": K * C, "note": "launch-bound at 1e4 elements"},
        "phases_ms": {"quantize_encrypt_xC": enc_ms, "aggregate_plus_decrypt_unquantize": float(ph[:, 1].mean())},
    })
    if world == 1 and not args.no_cpu_baseline:
        t0 = time.perf_counter()
        for _ in range(20):
            qq = [orc.quantize(xs[c], alpha, eb, us[c]) for c in range(C)]
            cc = [orc.encrypt(KEY, 0, c, "single", J, b, qq[c]) for c in range(C)]
            orc.unquantize(orc.decrypt(KEY, 0, [], list(range(C)), J, b, orc.aggregate_elem(cc, b)), alpha, eb, C)
        out["cpu_baseline"] = {"value": C * n / ((time.perf_counter() - t0) / 20), "unit": "ciphertexts/s", "cores": min(orc.num_threads(), usable_cpus()),
                               "kind": "port", "sample": "20 full rounds of the same workload through oracle/flashe_oracle.c"}
    return out


# ---- config 3: LeNet-sized model, 100 clients, double mask + mask precompute ---------------------------------------------
def bench_precompute(args, n, ops, rank, world, out):
    """Per round: every client's prepare_encrypt (the fused mask difference term(c) - term(c + 1), one chained launch for all
    clients), prepare_decrypt (term(C) - term(0)); then the ONLINE part with no AES at all: 100 x combine (ct = pt + mask),
    100-way reduce, combine (decrypt).  Both parts are inside the timed step; `phases_ms` splits them."""
    import numpy as np
    from oracle import flashe_oracle as orc          # parity gate only (rank 0, before the timed region)
    eng = ops.engine
    b, K, W, J = args.bits, args.steps, args.warmup, args.n_jobs
    L = 2 if b > 64 else 1
    C = args.clients or 100
    host_pts = [plaintext(c, n, b) for c in range(C)]
    pts = [ops.upload(p) for p in host_pts]
    masks = [eng.alloc_vec(n) for _ in range(C)]
    cts = [eng.alloc_vec(n) for _ in range(C)]
    dmask, agg, dec = eng.alloc_vec(n), eng.alloc_vec(n), eng.alloc_vec(n)
    ev = [[eng.event() for _ in range(4)] for _ in range(K)]
    fused_online = not args.no_fused_online
    # the per-round argument tables, built once: the same hundred clients' buffers go into every round's calls, and marshalling them
    # costs more host time than the launches take on the device (int_bits 23: the round was host-bound)
    t_jobs = eng.job_table([(c, c + 1, 0, n, None, 0, masks[c]) for c in range(C)] + [(C, 0, 0, n, None, 0, dmask)])
    t_pts, t_masks, t_cts = eng.ptr_table(pts), eng.ptr_table(masks), eng.ptr_table(cts)

    def step(it, k=None, split=False):
        """k: index of the timed step (events before / after the precompute launch, the dominant one); split: the untimed pass that
        also brackets the two online launches (phases_ms)."""
        if k is not None:
            eng.record(ev[k][0])
        # precompute: masks of every client (the chain shares the streams: C + 1 instead of 2 C) + the decrypt mask difference
        eng.prf_jobs_dev(it, n, J, t_jobs)
        if k is not None:
            eng.record(ev[k][1])
        if fused_online:
            # online encrypts ct = pt + (add - minus), the arbiter's reduce of them AND the decrypt of that reduce with the precomputed
            # decrypt mask from ONE pass (round 5: every ciphertext goes through the registers of the lane that owns the element; round
            # 6: a hundred clients without a minus operand are one launch, and the workgroup that completes an element's sum decrypts it)
            eng.combine_batch_sum_decrypt_dev(n, t_pts, 1, t_masks, None, t_cts, agg, dmask, None, dec)
            if split:
                eng.record(ev[k][2])
                eng.record(ev[k][3])
            return
        eng.combine_batch_dev(n, t_pts, 1, t_masks, None, t_cts)           # online encrypts, one launch
        if split:
            eng.record(ev[k][2])
        eng.aggregate_elem_dev(t_cts, n, agg)
        eng.combine_dev(n, agg, L, dmask, None, dec)                      # online decrypt
        if split:
            eng.record(ev[k][3])

    def poison():
        """Nothing a form of the round skipped may inherit a previous form's correct bytes: every output buffer is overwritten first."""
        for buf, pat in [(dec, 0x3C), (agg, 0xA5), (dmask, 0x69)] + [(c_, 0x5A) for c_ in cts] + [(m_, 0x96) for m_ in masks]:
            eng.memset_dev(buf, pat, buf.nbytes)

    poison()
    step(0)
    got = dec.download(np.uint64, n * L).reshape(n, L)
    lo, hi = sum_mod(host_pts, n, b)
    assert np.array_equal(got[:, 0], lo) and (L == 1 or np.array_equal(got[:, 1], hi)), "PARITY FAILURE (round trip)"
    orc.build()
    for c in (0, C // 2, C - 1):                       # ciphertexts against the oracle's own encrypt
        assert np.array_equal(cts[c].download(np.uint64, n * L).reshape(n, L), orc.encrypt(KEY, 0, c, "double", J, b, host_pts[c])), f"PARITY FAILURE client {c}"
    # the other form of the online half (two launches / one) beside the timed one, same buffers, parity-gated
    split_ms = None
    if fused_online:
        fused_online = False
        poison()
        step(0)
        got2 = dec.download(np.uint64, n * L).reshape(n, L)
        assert np.array_equal(got2, got), "PARITY FAILURE (split online half)"
        for c in (0, C // 2, C - 1):
            assert np.array_equal(cts[c].download(np.uint64, n * L).reshape(n, L), orc.encrypt(KEY, 0, c, "double", J, b, host_pts[c])), f"PARITY FAILURE client {c} (split online half)"
        settle(ops, step)
        for it in range(max(W, 3)):
            step(it)
        split_ms = timed_region(ops, K, lambda k: step(k)) * 1e3 / K
        fused_online = True
    settle(ops, step)
    for it in range(max(W, 3)):
        step(it)
    elapsed = timed_region(ops, K, lambda k: step(k, k))
    pre_ms = float(two_event_phases(eng, ev)[:, 0].mean())               # the precompute launch, HIP events inside the timed region
    for k in range(K):                                                    # the split of the online half: the same K rounds again, untimed
        step(k, k, split=True)
    ops.sync()
    ph = np.array([[eng.elapsed_ms(e[i], e[i + 1]) for i in range(3)] for e in ev])
    # The same K rounds as ONE graph launch each: the four launches of a round are captured once and replayed with the iter
    # shift advancing (flashe_graph_launch_shifted: every replay is a new round, no mask stream is reused).  Reported beside the
    # call-by-call figure; `value` stays the call-by-call one.
    graph_ms = None
    try:
        eng.graph_begin()
        step(0)
        graph = eng.graph_end()
        for k in range(3):
            graph.launch(iter_shift=k + 1)
        graph_ms = timed_region(ops, K, lambda k: graph.launch(iter_shift=k + 1)) * 1e3 / K
        got = dec.download(np.uint64, n * L).reshape(n, L)                   # the last replay = round K: same plaintexts, same sum
        assert np.array_equal(got[:, 0], lo) and (L == 1 or np.array_equal(got[:, 1], hi)), "PARITY FAILURE (graph replay)"
        assert np.array_equal(cts[1].download(np.uint64, n * L).reshape(n, L), orc.encrypt(KEY, K, 1, "double", J, b, host_pts[1])), \
            "PARITY FAILURE (graph replay, iter shift)"
    except AssertionError:
        raise
    except Exception as exc:
        print(f"graph replay unavailable: {exc!r}", file=sys.stderr)
    if rank != 0:
        return None
    m = 1 if L == 2 else 128 // b
    blocks = (C + 1 + 2) * ((n + m - 1) // m) if L == 2 else 2 * (C + 1) * ((n + m - 1) // m)
    alg_bytes = (C + 1) * n * 8 * L
    achieved = alg_bytes / (pre_ms * 1e-3) / 1e9
    out.update({
        "value": world * C * n / (elapsed / K), "ms_per_step": elapsed * 1e3 / K, "scaling": "weak",
        "config": {"workload": f"BASELINE config 3: LeNet-sized gradient (n={n}), {C} clients, double mask + mask precompute, {b}-bit modulus, "
                               f"n_jobs={J}; step = prepare_encrypt x {C} + prepare_decrypt (one launch) + online {C} encrypts + {C}-way "
                               "aggregate + decrypt" + (" (ONE pass, one launch)" if fused_online else "") + " (no AES online)" + ("; independent replicas per GPU" if world > 1 else ""),
                   "n": n, "int_bits": b, "clients_total": C, "mask": "double+precompute",
                   "parity": "bit-exact (round trip + three clients' ciphertexts vs the oracle, checked in-run)"},
        "roofline": {"kernel": "prf_chain_kernel<1024> (mask precompute: chain of %d clients + decrypt mask difference, in = NULL)" % C if L == 2
                     else "prf_small_chain_kernel<false> (mask precompute: chained streams, two streams per step)",
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": pre_ms, "launches_timed": K,
                     "aes_blocks_per_launch": blocks, "aes_blocks_per_s": blocks / (pre_ms * 1e-3),
                     "note": "launch- and latency-bound at this size (61,706 elements per vector)"},
        "phases_note": "precompute launch: HIP events inside the timed region (the only records in it); the online launches: an untimed pass over the same K rounds",
        "phases_ms": {"precompute_all_masks": pre_ms,
                      ("online_encrypt_xC_plus_aggregate_plus_decrypt" if fused_online else "online_encrypt_xC"): float(ph[:, 1].mean()),
                      ("online_decrypt_in_the_same_launch" if fused_online else "online_aggregate_plus_decrypt"): 0.0 if fused_online else float(ph[:, 2].mean()),
                      "ms_per_step_split_online_half": split_ms,
                      "round_as_one_graph_launch": graph_ms,
                      "graph_note": "the round's launches captured once and replayed with an advancing device-side iter shift (every "
                                    "replay is a new round); checked against the oracle at the last replayed iter.  A measurement, not a "
                                    "recommendation: a replay costs what the launches it replaces cost (DESIGN 4.4)"},
    })
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(J, b, C, host_pts, n)
    return out


# ---- config 2 at int_bits <= 32 in the compact layout (uint32 per element) --------------------------------------------------------
def bench_compact(args, n, ops, rank, world, out):
    """The config-2 round -- C encrypts in one chained launch, the C-way reduce fused with the decrypt of its result -- on uint32
    plaintext / ciphertext arrays (flashe_encrypt_batch_u32_dev, flashe_aggregate_decrypt_u32_dev): the same values as the one-limb
    layout, half the bytes.  Several GPUs: independent replicas."""
    import numpy as np
    from oracle import flashe_oracle as orc          # parity gate only (before the timed region)
    from flashe_amd.engine import SCHEME_DOUBLE
    eng = ops.engine
    b, K, W, J = args.bits, args.steps, args.warmup, args.n_jobs
    if b > 32:
        raise SystemExit("--layout u32 needs --bits <= 32")
    C = args.clients or 10
    host_pts = [plaintext(c, n, b) for c in range(C)]
    pts = [ops.upload(p.astype(np.uint32)) for p in host_pts]
    cts = [eng.alloc(4 * n + 16) for _ in range(C)]
    dec, agg = eng.alloc(4 * n + 16), eng.alloc(4 * n + 16)
    idx = list(range(C))
    ev = [[eng.event() for _ in range(2)] for _ in range(K)]
    # round 5: the encrypt launch also writes the local partial aggregate (the sum of its C ciphertexts), the second
    # launch decrypts that ONE vector -- the compact twin of the int_bits > 64 default; --no-partial-agg = encrypts, then reduce + decrypt
    partial = not args.no_partial_agg
    form = {"partial": partial}

    def step(it, k=None):
        if k is not None:
            eng.record(ev[k][0])
        if form["partial"]:
            eng.encrypt_batch_sum_u32_dev(it, idx, SCHEME_DOUBLE, n, J, pts, cts, agg)
        else:
            eng.encrypt_batch_u32_dev(it, idx, SCHEME_DOUBLE, n, J, pts, cts)
        if k is not None:
            eng.record(ev[k][1])
        eng.aggregate_decrypt_u32_dev(it, [C], [0], n, J, 0, n, [agg] if form["partial"] else cts, None, dec, 4)

    lo, _hi = sum_mod(host_pts, n, b)
    orc.build()
    other_ms = None
    for use_partial in ([not partial, partial]):              # the other form first (parity-gated, timed outside the reported region)
        form["partial"] = use_partial
        for buf, pat in [(dec, 0x3C), (agg, 0xA5)] + [(c_, 0x5A) for c_ in cts]:      # nothing a form skipped may inherit the other form's bytes
            eng.memset_dev(buf, pat, buf.nbytes)
        step(0)
        assert np.array_equal(dec.download(np.uint32, n).astype(np.uint64), lo), "PARITY FAILURE (round trip)"
        for c in (0, C - 1):
            assert np.array_equal(cts[c].download(np.uint32, n), orc.encrypt(KEY, 0, c, "double", J, b, host_pts[c])[:, 0].astype(np.uint32)), f"PARITY FAILURE client {c}"
        settle(ops, step)
        for it in range(max(W, 12)):
            step(it)
        if use_partial != partial:
            other_ms = timed_region(ops, K, lambda k: step(k)) * 1e3 / K
    elapsed = timed_region(ops, K, lambda k: step(k, k))
    if rank != 0:
        return None
    ph = two_event_phases(eng, ev)
    enc_ms = float(ph[:, 0].mean())
    m = 128 // b
    alg_bytes = C * n * (4 + 4) + (n * 4 if partial else 0)
    blocks = (C + 1) * ((n + m - 1) // m)
    achieved = alg_bytes / (enc_ms * 1e-3) / 1e9
    out.update({
        "value": world * C * n / (elapsed / K), "ms_per_step": elapsed * 1e3 / K, "scaling": "weak", "dtype": "u32",
        "config": {"workload": f"BASELINE config 2 at int_bits = {b}: n={n}-element vector, {C} clients per GPU, double mask, n_jobs={J}, uint32 "
                               "plaintext / ciphertext arrays (compact layout); round = one chained launch for the encrypts + the reduce fused with "
                               "the decrypt" + ("; independent replicas per GPU" if world > 1 else ""),
                   "n": n, "int_bits": b, "clients_total": C, "mask": "double", "layout": "uint32 per element",
                   "parity": "bit-exact (round trip + two clients' ciphertexts vs the oracle, checked in-run)"},
        "roofline": {"kernel": "prf_small_chain_kernel<PAIR, uint32> (chained encrypts, 4-byte elements)", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": enc_ms, "launches_timed": K,
                     "aes_blocks_per_launch": blocks, "aes_blocks_per_s": blocks / (enc_ms * 1e-3)},
        "phases_ms": {("encrypt_xC_plus_partial_aggregate" if partial else "encrypt_xC"): enc_ms,
                      ("decrypt_of_the_aggregate" if partial else "reduce_plus_decrypt"): float(ph[:, 1].mean())},
        ("ms_per_step_two_launch" if partial else "ms_per_step_partial_agg"): other_ms,
    })
    out["config"]["schedule_name"] = "partial-agg" if partial else "two-launch"
    return out


def cpu_baseline_sparse(n_jobs, b, C, locs, vals, zero, total):
    """The sparse round through the oracle on the host cores, in the reference's own structure: C compact single-mask encrypts, one
    expand_to_dense per upload + the running mod-add (jzf_aggregator.py:150-165, :419-430), the dense minus-mask from the location lists
    and its subtraction (jzf_flashe.py:316-343, :531-532).  Bounded: the expand + reduce is timed on the first uploads and scaled."""
    import numpy as np
    from oracle import flashe_oracle as orc
    orc.build()
    cores = min(orc.num_threads(), usable_cpus())
    orc.set_num_threads(cores)
    orc.mask(KEY, 0, 0, 1000, 1, b)          # table init outside the clock
    L = 2 if b > 64 else 1
    t0 = time.perf_counter()
    cts = [orc.encrypt(KEY, 0, c, "single", n_jobs, b, vals[c]) for c in range(C)]
    t1 = time.perf_counter()
    ns = min(C, 8)
    z = np.array([[zero] + [0] * (L - 1)], dtype=np.uint64)
    agg = np.zeros((total, L), dtype=np.uint64)
    for c in range(ns):
        agg = orc.aggregate_elem([agg, orc.expand_to_dense(total, locs[c], cts[c], z, b)], b, out=agg)
    t2 = time.perf_counter()
    mask = orc.sparse_minus_mask(KEY, 0, locs, total, n_jobs, b)
    orc.combine(b, agg, None, mask)
    t3 = time.perf_counter()
    t_round = (t1 - t0) + (t2 - t1) * C / ns + (t3 - t2)
    k = len(locs[0])
    return {"value": C * k / t_round, "unit": "ciphertexts/s", "cores": cores, "kind": "port", "ms_per_round": t_round * 1e3,
            "sample": f"one round through oracle/flashe_oracle.c: all {C} encrypts and the dense minus-mask + subtraction in full, expand_to_dense + "
                      f"running reduce timed on the first {ns} uploads and scaled to {C}",
            "phases_ms": {"encrypt_xC": (t1 - t0) * 1e3, f"expand_plus_reduce_x{C}_scaled": (t2 - t1) * C / ns * 1e3, "minus_mask_plus_decrypt": (t3 - t2) * 1e3}}


# ---- config 5: top-1 % sparse uploads, 50 clients, single mask (the sparse path the reference can run) --------------------
def bench_sparse(args, total, ops, rank, world, out):
    """Per round: every client encrypts its compact k-vector (single mask, compact positions);
    the arbiter adds the 50 expanded uploads (fused: sum of zero values everywhere + (value - zero) at the locations); the
    clients' dense minus-mask is rebuilt from the location lists and subtracted."""
    import numpy as np
    from oracle import flashe_oracle as orc          # parity gate only
    from flashe_amd.engine import SCHEME_SINGLE
    eng = ops.engine
    b, K, W, J = args.bits, args.steps, args.warmup, args.n_jobs
    L = 2 if b > 64 else 1
    C = args.clients or 50
    k = total // 100
    rng = [np.random.Generator(np.random.PCG64(2000 + c)) for c in range(C)]
    locs = [np.sort(r.choice(total, k, replace=False)).astype(np.uint32) for r in rng]
    vals = [r.integers(0, 2 ** 64 if b >= 64 else 2 ** (b - 8), k, dtype=np.uint64) for r in rng]
    zero = 1 << 31                                   # the un-encrypted quantised zero that closes every upload
    d_loc = [ops.upload(l) for l in locs]
    d_val = [ops.upload(v) for v in vals]
    d_ct = [eng.alloc_vec(k) for _ in range(C)]
    d_agg, d_dec = eng.alloc_vec(total), eng.alloc_vec(total)
    ev = [[eng.event() for _ in range(4)] for _ in range(K)]
    # the span bounds of the round's location lists (where every list enters every span of the dense vector): computed ONCE per round
    # and shared by the sparse aggregate and the sparse decrypt (round 4; both used to run that pass on the same lists)
    use_bounds = L == 2 and not args.no_span_bounds
    bounds = None

    fused_ok = L == 2 and not args.sparse_separate
    # the per-round argument tables, built once: the same 50 clients' buffers go into every round's calls, and building the ctypes arrays
    # costs about as much host time as the launches they describe take on the device
    t_loc, t_val, t_ct, t_k = eng.ptr_table(d_loc), eng.ptr_table(d_val), eng.ptr_table(d_ct), eng.u64_table([k] * C)
    t_zero, idx_all = eng.zeros_table([zero] * C), list(range(C))

    def rec(kk, i, dom, split):
        """Timed steps record the two events around the dominant launch only (an event record costs ~3 us on the stream); the split
        pass after the timed region records all four."""
        if kk is not None and (split or i in dom):
            eng.record(ev[kk][i])

    def step_separate(it, kk=None, split=False):
        """Round-2 .. 4 form: the encrypts (one chained launch), the arbiter's sparse aggregate, the sparse decrypt."""
        rec(kk, 0, (0, 1), split)
        eng.encrypt_batch_dev(it, idx_all, SCHEME_SINGLE, k, J, t_val, 1, t_ct)
        rec(kk, 1, (0, 1), split)
        if bounds is not None:
            bounds.recompute(t_loc, t_k)             # a real job has new lists every round: the pass is part of the step
        eng.sparse_aggregate_dev(total, t_loc, t_k, t_ct, t_zero, d_agg, sorted_lists=True, bounds=bounds)
        rec(kk, 2, (0, 1), split)
        eng.sparse_decrypt_dev(it, t_loc, t_k, total, J, d_agg, d_dec, sorted_lists=True, bounds=bounds)    # dense minus-mask built and subtracted in one pass
        rec(kk, 3, (0, 1), split)

    def step_fused(it, kk=None, split=False):
        """The clients this GPU plays encrypt AND their uploads are summed in one persistent launch (flashe_sparse_encrypt_aggregate_dev,
        the sparse twin of the dense round's partial aggregate); the ciphertexts are still written, the decrypt is the other party's pass."""
        rec(kk, 0, (1, 2), split)
        if bounds is not None:
            bounds.recompute(t_loc, t_k)
        rec(kk, 1, (1, 2), split)
        eng.sparse_encrypt_aggregate_dev(it, idx_all, t_loc, t_k, t_val, 1, t_zero, total, J, t_ct, d_agg, bounds=bounds)
        rec(kk, 2, (1, 2), split)
        eng.sparse_decrypt_dev(it, t_loc, t_k, total, J, d_agg, d_dec, sorted_lists=True, bounds=bounds)
        rec(kk, 3, (1, 2), split)

    bounds = eng.span_bounds(total, t_loc, t_k) if use_bounds else None
    step = step_fused if fused_ok else step_separate
    orc.build()
    # the expected dense result as FULL-WIDTH integers (both limbs: with 64-bit values and 50 clients nearly every touched position carries
    # into the high limb): (C - #clients holding p) * zero + the 128-bit sum of the values uploaded for p, mod 2^b
    want = np.zeros((total, L), dtype=np.uint64)
    held = np.zeros(total, dtype=np.uint64)
    for c in range(C):
        old = want[locs[c], 0]
        new = old + vals[c]
        want[locs[c], 0] = new
        if L == 2:
            want[locs[c], 1] += (new < old).astype(np.uint64)
        held[locs[c]] += np.uint64(1)
    rest = (np.uint64(C) - held) * np.uint64(zero)
    new = want[:, 0] + rest
    if L == 2:
        want[:, 1] += (new < want[:, 0]).astype(np.uint64)
    want[:, 0] = new
    if b % 64:
        want[:, L - 1] &= np.uint64((1 << (b % 64)) - 1)
    del held, rest, new
    for st in ([step_fused, step_separate] if fused_ok else [step_separate]):
        for buf, pat in [(d_dec, 0x3C), (d_agg, 0xA5)] + [(c_, 0x5A) for c_ in d_ct]:  # nothing a schedule skipped may inherit the other one's bytes
            eng.memset_dev(buf, pat, buf.nbytes)
        st(0)
        # parity: the decrypted dense vector == sum over clients of (value at its locations, zero elsewhere)
        got = d_dec.download(np.uint64, total * L).reshape(total, L)
        assert np.array_equal(got, want), "PARITY FAILURE (sparse round trip, every limb)"
        for c in (3, C - 1):
            assert np.array_equal(d_ct[c].download(np.uint64, k * L).reshape(k, L), orc.encrypt(KEY, 0, c, "single", J, b, vals[c])), f"PARITY FAILURE client {c}"
    sep_ms = None
    if fused_ok:                                         # the other schedule beside it, same buffers
        settle(ops, step_separate)
        for it in range(max(W, 2)):
            step_separate(it)
        sep_ms = timed_region(ops, K, lambda kk: step_separate(kk)) * 1e3 / K
    settle(ops, step)
    for it in range(max(W, 2)):
        step(it)
    elapsed = timed_region(ops, K, lambda kk: step(kk, kk))
    d0 = 1 if fused_ok else 0
    dom_ms = float(np.mean([eng.elapsed_ms(e[d0], e[d0 + 1]) for e in ev]))   # the dominant launch, HIP events inside the timed region
    for kk in range(K):                                   # the split of the rest of the round: the same K rounds again, untimed, every event
        step(kk, kk, split=True)
    ops.sync()
    # several GPUs: `value` above is every rank running the whole round on its own data (replicas, weak scaling).  Beside it, ONE round
    # shared by all ranks: the dense vector cut into position ranges of whole spans, every rank plays every client on the range it owns
    # (SparseShardedRound: no exchange for the aggregate, the decrypted ranges all-gathered) -- strong scaling of the same 50-client round.
    shard_ms = shard_ms_no_gather = None
    if world > 1 and L == 2 and not args.no_position_sharded:
        from flashe_amd.dist import SparseShardedRound
        srnd = SparseShardedRound(ops, total, b, C, J, rank=rank, world=world)
        refs = lambda bufs: [(bf, 0) for bf in bufs]                      # noqa: E731
        r_loc, r_val, r_ct = refs(d_loc), refs(d_val), refs(d_ct)
        out_ref = srnd.run(0, r_loc, [k] * C, r_val, 1, [zero] * C, r_ct)
        got = ops.read((out_ref, 0), total * L).reshape(total, L)
        assert np.array_equal(got, want), "PARITY FAILURE (position-sharded sparse round, every limb)"
        for it in range(max(W, 2)):
            srnd.run(it, r_loc, [k] * C, r_val, 1, [zero] * C, r_ct)
        shard_ms = timed_region(ops, K, lambda kk: srnd.run(kk, r_loc, [k] * C, r_val, 1, [zero] * C, r_ct)) * 1e3 / K
        shard_ms_no_gather = timed_region(ops, K, lambda kk: srnd.run(kk, r_loc, [k] * C, r_val, 1, [zero] * C, r_ct, gather=False)) * 1e3 / K
    if rank != 0:
        return None
    if shard_ms is not None:
        out.update({"value_position_sharded": C * k / (shard_ms * 1e-3), "ms_per_step_position_sharded": shard_ms,
                    "value_position_sharded_no_gather": C * k / (shard_ms_no_gather * 1e-3),
                    "position_sharded_note": f"ONE {C}-client round over {world} GPUs: position ranges of whole spans, no exchange for the "
                                             "aggregate; all-gather of the decrypted ranges included in the first figure (strong scaling)"})
    ph = np.array([[eng.elapsed_ms(e[i], e[i + 1]) for i in range(3)] for e in ev])
    m = 1 if L == 2 else 128 // b
    prf_blocks = C * ((k + m - 1) // m)
    if fused_ok:
        enc_ms, dec_ms = dom_ms, float(ph[:, 2].mean())
        enc_bytes = C * k * (4 + 8 + 8 * L) + total * 8 * L          # locations + plaintexts + ciphertexts, the dense aggregate written once
        dec_bytes = C * k * 4 + 2 * total * 8 * L                     # locations, the dense aggregate read and the result written
        achieved = enc_bytes / (enc_ms * 1e-3) / 1e9
        ratio, tsrc = traffic_ratio("span_prf_kernel")
        dratio, _ = traffic_ratio("span_prf_kernel_decrypt")
        out.update({
            "value": world * C * k / (elapsed / K), "ms_per_step": elapsed * 1e3 / K, "scaling": "weak",
            "value_separate_launches": world * C * k / (sep_ms * 1e-3), "ms_per_step_separate_launches": sep_ms,
            "config": {"workload": f"BASELINE config 5: top-1 % sparsified gradient (k={k} of {total} positions, u32 index + {8 * L}-byte value), {C} clients, "
                                   f"{b}-bit modulus, single mask over compact positions (the sparse path the reference runs; dynamic masking picks it), "
                                   f"n_jobs={J}; step = span bounds of the round's lists + {C} compact encrypts with the sum of the expanded uploads written "
                                   "in the same persistent launch + sparse decrypt (mask blocks computed inside the span reduce)"
                                   + ("; independent replicas per GPU" if world > 1 else ""),
                       "n": total, "k": k, "int_bits": b, "clients_total": C, "mask": "single (sparse)", "schedule": "encrypt+aggregate fused",
                       "parity": "bit-exact (dense round trip + two clients' ciphertexts vs the oracle, both schedules, checked in-run)"},
            "roofline": {"kernel": f"span_prf_kernel<1> ({C} clients' single-mask encrypts over compact positions + the sparse aggregate of their uploads: one "
                                   "persistent launch, one workgroup per CU, AES tables and span accumulators in the same 160 KiB of LDS)",
                         "kernel_key": "span_prf_kernel", "bound": "lds", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": ratio * enc_bytes if ratio else None, "traffic_source": tsrc if ratio else None,
                         "algorithmic_bytes_per_launch": enc_bytes, "avg_launch_ms": enc_ms, "launches_timed": K, "aes_blocks_per_launch": prf_blocks, "aes_blocks_per_s": prf_blocks / (enc_ms * 1e-3),
                         "note": "the round's two dominant launches are this one and its decrypt twin; one AES block per list entry, LDS-lookup / "
                                 "VALU-issue bound, `frac` is its HBM fraction as required"},
            "roofline_sparse_decrypt": {"kernel": "span_prf_kernel<0> (dense minus-mask built inside the span reduce and subtracted from the aggregate)",
                                        "bound": "lds", "achieved": dec_bytes / (dec_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": dec_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": dratio * dec_bytes if dratio else None,
                                        "algorithmic_bytes_per_launch": dec_bytes,
                                        "avg_launch_ms": dec_ms, "aes_blocks_per_s": prf_blocks / (dec_ms * 1e-3)},
            "span_bounds": "computed once per round (both span sizes in one pass), shared by the two passes" if bounds is not None else "computed by each pass",
            "phases_note": "encrypt launch: HIP events inside the timed region (the only records in it); bounds and decrypt: an untimed pass over the same K rounds",
            "phases_ms": {"span_bounds": float(ph[:, 0].mean()), "encrypt_xC_plus_sparse_aggregate": enc_ms, "minus_mask_plus_decrypt": dec_ms},
        })
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sparse(J, b, C, locs, vals, zero, total)
        return out
    agg_ms = float(ph[:, 1].mean())
    enc_ms = dom_ms
    alg_bytes = C * k * (4 + 8 * L) + total * 8 * L
    achieved = alg_bytes / (agg_ms * 1e-3) / 1e9
    # the launch that dominates the round BY TIME is the PRF chain of the compact single-mask streams
    prf_bytes = C * k * (8 + 8 * L)
    prf_achieved = prf_bytes / (enc_ms * 1e-3) / 1e9
    out.update({
        "value": world * C * k / (elapsed / K), "ms_per_step": elapsed * 1e3 / K, "scaling": "weak",
        "config": {"workload": f"BASELINE config 5: top-1 % sparsified gradient (k={k} of {total} positions, u32 index + {8 * L}-byte value), {C} clients, "
                               f"{b}-bit modulus, single mask over compact positions (the sparse path the reference runs; dynamic masking picks it), "
                               f"n_jobs={J}; step = {C} compact encrypts + fused sparse aggregate + decrypt (dense minus-mask built and subtracted in one pass)"
                               + ("; independent replicas per GPU" if world > 1 else ""),
                   "n": total, "k": k, "int_bits": b, "clients_total": C, "mask": "single (sparse)", "schedule": "separate launches",
                   "parity": "bit-exact (dense round trip + two clients' ciphertexts vs the oracle, checked in-run)"},
        "roofline": {"kernel": (f"prf_chain_kernel<1024> ({C} single-mask streams over compact positions: the {C} clients' encrypts in one launch)") if L == 2 else
                               f"prf_small_chain_kernel ({C} single-mask streams over compact positions)",
                     "kernel_key": "prf_chain_kernel" if L == 2 else "prf_small_chain_kernel",
                     "bound": "lds", "achieved": prf_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": prf_achieved / HBM_PEAK_GBPS,
                     "traffic": None, "algorithmic_bytes_per_launch": prf_bytes, "avg_launch_ms": enc_ms, "launches_timed": K,
                     "aes_blocks_per_launch": prf_blocks, "aes_blocks_per_s": prf_blocks / (enc_ms * 1e-3),
                     "note": "AES-rate (LDS lookup) bound, `frac` is its HBM fraction as required"},
        "roofline_sparse_aggregate": {"kernel": "span_reduce_kernel (fused sparse aggregate: LDS-staged spans, dense vector written once)",
                                      "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                                      "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": agg_ms,
                                      "note": "phase = span bounds of the round's lists (once, shared with the decrypt) + the span reduce"},
        "span_bounds": "computed once per round, shared by aggregate and decrypt" if bounds is not None else "computed by each of the two passes",
        "phases_ms": {"encrypt_xC": enc_ms, "sparse_aggregate": agg_ms, "minus_mask_plus_decrypt": float(ph[:, 2].mean())},
    })
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_sparse(J, b, C, locs, vals, zero, total)
    return out


if __name__ == "__main__":
    main()
