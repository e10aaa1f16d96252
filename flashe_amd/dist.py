"""Multi-GPU round: client shards over the GPUs of one node, one process per GPU -- no PyTorch.

Mapping of the reference's application-level collectives: the arbiter's
*gather* of client models + *reduce* in Python (jzf_aggregator.py:292-308, :404-430) + *broadcast*
of the aggregate (:502-508) become, inside one node:

  1. every rank encrypts its own clients' vectors and mod-adds them locally (no exchange);
  2. ONE exchange step: a reduce-scatter of the per-rank partial aggregates.  RCCL has no 128-bit
     integer type, so it is an all-to-all of slices (rank g receives slice g of every partial; on
     xGMI every GPU pair has its own link, so all 7 links run concurrently instead of a
     per-link-bound ring: flashe_rccl_all_to_all) followed by the local C-way mod-add kernel;
  3. every rank decrypts the slice it owns (PRF counters are position-indexed:
     flashe_decrypt_range_dev) and an all-gather hands the plaintext aggregate to all ranks.

`run_packed` is the same round with the arbiter's PACKED reduce (one n*b-bit integer per model,
carries crossing element boundaries): limb slices instead of element slices plus one small
all-gather of per-slice carry information, resolved on the device.

Everything goes through the C ABI: kernels, device memory, and the collectives (`flashe_rccl_*`,
RCCL loaded by libflashe_hip.so itself).  The schedules are written against an `ops` object --
`HipOps` in production; the CPU tests substitute a host double with the same surface (under
tests/, gloo for the exchange).  A *ref* is `(buffer, offset in uint64 words)`.
"""
import os
import subprocess
import sys
import time

import numpy as np

from . import _lib
from .engine import SCHEME_DOUBLE, Engine, limbs_of, telescope

__all__ = ["HipOps", "RcclComm", "ShardedRound", "slice_len", "deal_clients", "spawn", "rendezvous_unique_id", "Watchdog", "run_tag"]

ALIGN = 256         # slice and chunk boundaries are multiples of this many elements (256 consecutive PRF counters share 3 bytes)


def slice_len(n, world):
    per = (n + world - 1) // world
    return ((per + ALIGN - 1) // ALIGN) * ALIGN


def deal_clients(total, world):
    """Global client numbers hosted by every rank: contiguous blocks, the first `total % world` ranks one client more
    (10 clients on 8 GPUs -> 2, 2, 1, 1, 1, 1, 1, 1 as BASELINE config 4 deals them).  Blocks rather than round-robin: the same
    counts per rank, and consecutive clients of one rank share their PRF streams (flashe_encrypt_batch_dev)."""
    base, extra = divmod(total, world)
    out, c = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(list(range(c, c + k)))
        c += k
    return out


# ------------------------------------------------------------------------------------------------------------------
# process launch and rendezvous (no GPU call happens before the per-GPU processes exist)
# ------------------------------------------------------------------------------------------------------------------
def _resolve_prctl():
    """libc's prctl, looked up ONCE at import: the preexec hook below runs between fork and exec, where an `import` or a dlopen can
    deadlock on a lock another thread of the launcher held at the fork (the Python docs' warning about preexec_fn)."""
    try:
        import ctypes
        fn = ctypes.CDLL("libc.so.6", use_errno=True).prctl
        fn.argtypes = [ctypes.c_int, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong]
        fn.restype = ctypes.c_int
        return fn
    except Exception:
        return None


_PRCTL = _resolve_prctl()
_SIGKILL = 9


def _die_with_parent():
    """preexec hook of a rank process (runs between fork and exec, before anything GPU): the kernel sends SIGKILL to the rank
    when the launcher dies, so that even a `kill -9` of the launcher leaves no rank behind holding a GPU.  One C call, nothing
    imported or loaded here.  PR_SET_PDEATHSIG is tied to the THREAD that forked: `spawn` must be called from a thread that outlives
    the ranks (the main thread; `spawn` itself blocks until they are done, so calling it and waiting in the same thread is enough)."""
    if _PRCTL is not None:
        _PRCTL(1, _SIGKILL, 0, 0, 0)                   # PR_SET_PDEATHSIG


def spawn(n_procs, argv, master_port=None, env=None, deadline_s=None):
    """Start `n_procs` processes `python argv...`, one per GPU, with the usual RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT environment, and wait for them.  Returns the first non-zero exit code (0 if none).
    Call it BEFORE anything in the parent touches the GPU.

    All ranks are watched together: as soon as one exits non-zero (parity failure, OOM, ncclCommInitRank error) the others are
    terminated -- RCCL has no timeout, a surviving rank would sit in its next collective forever, holding its GPU.  SIGTERM / SIGINT
    of the launcher are forwarded, every rank runs in its own session (killpg reaches what it started) and is killed by the kernel if
    the launcher itself is killed.  deadline_s: optional overall limit (exit code 124, like timeout(1)).  Never re-execs."""
    import signal
    port = int(master_port or (29400 + os.getpid() % 500))
    run_id = f"{os.getpid()}_{int(time.time() * 1e3)}"
    procs = []
    for r in range(n_procs):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_procs), LOCAL_WORLD_SIZE=str(n_procs),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FLASHE_RUN_ID=run_id)
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, start_new_session=True, preexec_fn=_die_with_parent))

    def stop_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    got = []
    previous = {}
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            previous[sig] = signal.signal(sig, lambda s, _f: got.append(s))
        except ValueError:                   # not the main thread: no handlers, the rest still works
            pass
    rc, t0, killing_since = 0, time.time(), None
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                code = p.poll()
                if code is not None and code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
            if got and rc == 0:
                rc = 128 + got[0]
            if deadline_s is not None and time.time() - t0 > deadline_s and rc == 0:
                rc = 124
            if rc and killing_since is None:
                killing_since = time.time()
                stop_all(signal.SIGTERM)
            elif killing_since is not None and time.time() - killing_since > 5.0:
                stop_all(signal.SIGKILL)
            time.sleep(0.05)
        for p in procs:
            code = p.returncode
            if code and rc == 0:
                rc = code if code > 0 else 128 - code
    finally:
        stop_all(signal.SIGKILL)
        for sig, h in previous.items():
            signal.signal(sig, h)
    return rc


def run_tag(world):
    """What names ONE launch on this node: every rank of it is a child of the same launcher process (torchrun's agent, or `spawn`),
    whose pid keeps launches that reuse a port apart, so a file left behind by a crashed run is never mistaken for this run's."""
    return "_".join([os.environ.get("MASTER_PORT", "0"),
                     os.environ.get("FLASHE_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                     os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), str(os.getppid()), str(world)])


def rdzv_dir():
    import tempfile
    return os.environ.get("FLASHE_RDZV_DIR", tempfile.gettempdir())


class Watchdog:
    """A deadline and an out-of-band abort channel for the ranks of one launch (RCCL collectives have no timeout: a rank that waits
    for a peer that raised, died or took another branch waits forever).

    Every rank runs one: a daemon thread that fires when the armed deadline passes or when ANY rank of the launch has called
    `abort(reason)` (a file next to the rendezvous file, polled).  Firing calls `on_fire(reason)` -- e.g. rank 0 prints the result it
    already holds -- and then leaves with os._exit: an exit, never an exec (a process that has touched the GPU must not be replaced
    in place), and no interpreter teardown that could block on the stuck stream.  ctypes calls release the GIL, so the thread runs
    while the main thread sits inside a collective."""

    def __init__(self, rank, world, on_fire=None, poll_s=0.1):
        import threading
        self.rank, self.world, self.on_fire = rank, world, on_fire
        self.exit_code = 3                      # what a firing exits with; callers lower it to 0 once a valid result is in hand
        # The flag lives in a directory only this user can write to (0700, ownership checked): in a shared /tmp a predictable flag NAME
        # could be squatted by another user -- the ranks would then find "a flag" they must not trust, nobody would fire, and the peers
        # of an aborting rank would sit in their collective until the deadline.  When the directory cannot be had (someone else owns
        # the name), abort() still fires locally and the peers leave at their deadline.
        self.dir = self._private_dir(os.path.join(rdzv_dir(), f"flashe_run_{os.getuid()}_{run_tag(world)}"))
        self.path = os.path.join(self.dir or rdzv_dir(), f"flashe_abort_{os.getuid()}_{run_tag(world)}")
        self._deadline, self._phase = None, ""
        self._lock = threading.Lock()
        self._done = False
        self._poll = poll_s
        self._thread = threading.Thread(target=self._watch, name="flashe-watchdog", daemon=True)
        self._thread.start()

    @staticmethod
    def _private_dir(path):
        """`path` as a directory of this user with mode 0700 (created if missing); None when the name is taken by anything else."""
        import stat
        try:
            os.mkdir(path, 0o700)
        except FileExistsError:
            pass
        except OSError:
            return None
        try:
            st = os.lstat(path)
        except OSError:
            return None
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            return None
        return path

    def arm(self, seconds, phase):
        self._deadline, self._phase = time.time() + float(seconds), phase

    def disarm(self):
        self._deadline = None

    def abort(self, reason):
        """Tell every rank of the launch (this one included) to stop: the watchdogs fire within a poll interval."""
        import tempfile
        tmp = None
        try:
            # the flag appears WITH its text or not at all (a watcher that found an empty file fired with no reason): written under a
            # private name, then hard-linked to the flag's name -- link() is atomic and fails when another rank's flag is already there
            fd, tmp = tempfile.mkstemp(prefix="flashe_abort_", dir=os.path.dirname(self.path))          # 0600
            with os.fdopen(fd, "w") as f:
                f.write(f"rank {self.rank}: {reason}")
            os.link(tmp, self.path)
        except FileExistsError:
            # another rank's flag is already there -- if it really is one of ours; a file of another user (or a stale one nobody of this
            # launch can read) will never make a watchdog fire: then at least this rank leaves now
            if not self._flag_is_ours():
                self._fire(f"rank {self.rank}: {reason} (the abort flag name is taken by a foreign file)")
        except OSError:
            # (no link() on this filesystem, or the directory is gone: only this rank fires; its peers leave at their deadline)
            self._fire(f"rank {self.rank}: {reason}")
        finally:
            if tmp is not None:
                try:
                    os.unlink(tmp)
                except OSError:
                    pass

    def _read_flag(self):
        """(text, ours?) of the flag file, from ONE open: ownership is checked on the descriptor that is read, not on the name."""
        try:
            fd = os.open(self.path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        except OSError:
            return None, False
        try:
            import stat
            st = os.fstat(fd)
            ours = st.st_uid == os.getuid() and stat.S_ISREG(st.st_mode)
            text = os.read(fd, 4096).decode(errors="replace") if ours else ""
        except OSError:
            text, ours = None, False
        finally:
            os.close(fd)
        return text, ours

    def _flag_is_ours(self):
        return self._read_flag()[1]

    def finish(self):
        """The normal end: from here on the watchdog never fires (returns False if it already has)."""
        with self._lock:
            if self._done:
                return False
            self._done = True
        if self.rank == 0:
            for rm, target in ((os.unlink, self.path), (os.rmdir, self.dir)):
                try:
                    if target:
                        rm(target)
                except OSError:
                    pass
        return True

    def _fire(self, reason):
        with self._lock:
            if self._done:
                return
            self._done = True
            try:
                if self.on_fire:
                    self.on_fire(reason)
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(self.exit_code)

    def _watch(self):
        while not self._done:
            time.sleep(self._poll)
            d = self._deadline
            if d is not None and time.time() > d:
                self._fire(f"deadline of phase '{self._phase}' passed on rank {self.rank}")
            why, ours = self._read_flag()
            if ours:
                self._fire(why or "abort requested")


def rendezvous_unique_id(rank, world, make_id, timeout=300.0):
    """Hand rank 0's 128-byte ncclUniqueId to every rank of the node through a file in a directory all of them see
    (FLASHE_RDZV_DIR, default the system temp dir), keyed by MASTER_PORT and the launcher's run id -- the environment
    torchrun or `spawn` already provides; no sockets, no PyTorch."""
    import tempfile
    d = rdzv_dir()
    path = os.path.join(d, f"flashe_rccl_id_{os.getuid()}_{run_tag(world)}")
    if rank == 0:
        ident = bytes(make_id())
        fd, tmp = tempfile.mkstemp(prefix="flashe_rccl_id_", dir=d)          # 0600, a name nobody else can have prepared
        with os.fdopen(fd, "wb") as f:
            f.write(ident)
        os.replace(tmp, path)                                   # atomic: readers see nothing or all 128 bytes
        return ident, path
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                ident = f.read()
                mine = os.fstat(f.fileno()).st_uid == os.getuid()          # only a file this user wrote is trusted
            if len(ident) == 128 and mine:
                return ident, path
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout} s")
        time.sleep(0.01)


class RcclComm:
    """flashe_comm: the RCCL communicator of this process (one GPU), created through the C ABI."""

    MAX, MIN, SUM = 0, 1, 2

    def __init__(self, engine, rank, world, unique_id):
        import ctypes
        self._lib = _lib.load()
        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        ident = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        engine._check(self._lib.flashe_rccl_init(engine._h, ident, rank, world, ctypes.byref(h)))
        self._h = h.value

    @staticmethod
    def new_unique_id():
        import ctypes
        lib = _lib.load()
        ident = (ctypes.c_uint8 * 128)()
        rc = lib.flashe_rccl_unique_id(ident)
        if rc:
            msg = lib.flashe_last_error(None)
            raise _lib.FlasheError(rc, msg.decode() if msg else "flashe_rccl_unique_id failed")
        return bytes(ident)

    @classmethod
    def from_env(cls, engine):
        """Rank / world from RANK and WORLD_SIZE (torchrun's or `spawn`'s environment); the unique id travels by file."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        ident, path = rendezvous_unique_id(rank, world, cls.new_unique_id)
        comm = cls(engine, rank, world, ident)                 # collective: every rank has read the file once this returns
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    def all_to_all(self, engine, send_ptr, send_stride, recv_ptr, recv_stride, nbytes):
        engine._check(self._lib.flashe_rccl_all_to_all(engine._h, self._h, send_ptr, send_stride, recv_ptr, recv_stride, nbytes))

    def all_gather(self, engine, send_ptr, recv_ptr, nbytes):
        engine._check(self._lib.flashe_rccl_all_gather(engine._h, self._h, send_ptr, recv_ptr, nbytes))

    def allreduce_modadd(self, engine, ptr, count):
        """int_bits <= 64: buf[j] = (sum over ranks of buf[j]) mod 2^b on every rank, in place (ncclAllReduce(ncclUint64, ncclSum) + mask)."""
        engine._check(self._lib.flashe_rccl_allreduce_modadd_u64(engine._h, self._h, ptr, int(count)))

    def allreduce(self, engine, value, op=0):
        import ctypes
        v = ctypes.c_double(float(value))
        engine._check(self._lib.flashe_rccl_allreduce_f64(engine._h, self._h, ctypes.byref(v), op))
        return float(v.value)

    def barrier(self, engine):
        engine._check(self._lib.flashe_rccl_barrier(engine._h, self._h))

    def rccl_world(self):
        """The number of ranks RCCL itself reports for this communicator (ncclCommCount through flashe_rccl_world)."""
        return int(self._lib.flashe_rccl_world(self._h))

    def close(self):
        if self._h is not None:
            self._lib.flashe_rccl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------------
# ops: what a schedule needs from the device
# ------------------------------------------------------------------------------------------------------------------
class HipOps:
    """Refs -> C-ABI `_dev` calls.  `side` is an optional second engine (same key / int_bits, its own stream) on which the
    HBM-bound reduce and the exchange of the pipelined schedules run next to the AES-bound kernels; `comm` an RcclComm
    (None: single rank without an exchange)."""

    def __init__(self, engine, side=None, comm=None):
        self.engine, self.side, self.comm = engine, side, comm
        self.rank = comm.rank if comm else 0
        self.world = comm.world if comm else 1
        self.L = engine.limbs
        self._events = {}

    # ---- memory ----
    def alloc(self, words):
        buf = self.engine.alloc(max(int(words) * 8, 16))
        self.engine._check(self.engine._lib.flashe_memset_dev(self.engine._h, buf.ptr, 0, buf.nbytes))
        return buf

    def upload(self, arr):
        return self.engine.upload(np.ascontiguousarray(arr))

    def read(self, ref, words):
        """Host copy of `words` uint64 words at ref (synchronises both streams)."""
        if self.side is not None:
            self.side.sync()
        out = np.empty(int(words), dtype=np.uint64)
        if words:
            self.engine._check(self.engine._lib.flashe_memcpy_d2h(self.engine._h, out.ctypes.data, self._a(ref), out.nbytes))
        else:
            self.engine.sync()
        return out

    def sync(self):
        self.engine.sync()
        if self.side is not None:
            self.side.sync()

    @staticmethod
    def _a(ref):
        if ref is None:
            return None
        buf, off = ref
        return (buf.ptr if hasattr(buf, "ptr") else int(buf)) + 8 * int(off)

    def _eng(self, side):
        return self.side if (side and self.side is not None) else self.engine

    # ---- two-stream plumbing of the pipelined schedules (no-ops without a side engine) ----
    def _ev(self, name):
        if name not in self._events:
            self._events[name] = self.engine.event()
        return self._events[name]

    def signal(self, name, side=False):
        if self.side is not None:
            self._eng(side).record(self._ev(name))

    def wait(self, name, side=False):
        if self.side is not None:
            self._eng(side).wait_event(self._ev(name))

    # ---- cipher ----
    def encrypt_batch(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts, sum_out=None):
        """sum_out: optional ref that receives the sum of the ciphertexts (the local partial aggregate) from the same launch."""
        if sum_out is not None:
            self.engine.encrypt_batch_sum_dev(it, idx_list, scheme, n, n_jobs, [self._a(r) for r in pts], pt_limbs,
                                              [self._a(r) for r in cts], self._a(sum_out))
        else:
            self.engine.encrypt_batch_dev(it, idx_list, scheme, n, n_jobs, [self._a(r) for r in pts], pt_limbs, [self._a(r) for r in cts])

    def encrypt_range(self, it, idx, scheme, n, n_jobs, first, count, pt, pt_limbs, ct):
        """pt / ct address element `first`."""
        self.engine.encrypt_range_dev(it, idx, scheme, n, n_jobs, first, count, self._a(pt), pt_limbs, self._a(ct))

    def encrypt_batch_range(self, it, idx_list, scheme, n, n_jobs, first, count, pts, pt_limbs, cts, sum_out=None):
        """Every listed client's encrypt on elements [first, first + count) (refs address element `first`), optionally with the slice
        of their sum from the same launch: what a rank that owns an element slice of every client vector runs."""
        self.engine.encrypt_batch_range_dev(it, idx_list, scheme, n, n_jobs, first, count, [self._a(r) for r in pts], pt_limbs,
                                            [self._a(r) for r in cts], self._a(sum_out))

    class _View:
        """`words`-offset window into a device buffer (keeps the buffer alive): lets a schedule hand out a result that does not start
        at the beginning of its allocation as a plain buffer."""

        def __init__(self, buf, off_words):
            self.parent, self.ptr = buf, buf.ptr + 8 * int(off_words)

    def view(self, buf, off_words):
        return HipOps._View(buf, off_words) if off_words else buf

    def decrypt_range(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out, side=False):
        self._eng(side).decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, self._a(inp), self._a(out))

    def aggregate(self, srcs, count, out, side=False):
        self._eng(side).aggregate_elem_dev([self._a(r) for r in srcs], count, self._a(out))

    def aggregate_decrypt(self, it, add_idx, minus_idx, n, n_jobs, first, count, srcs, agg_out, out, side=False):
        """The reduce fused with the decrypt of its result on elements [first, first + count); refs address element `first`."""
        self._eng(side).aggregate_decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, [self._a(r) for r in srcs],
                                                    self._a(agg_out), self._a(out))

    def prf_jobs(self, it, n, n_jobs, jobs):
        """jobs: (add_idx, minus_idx or None, first, count, in_ref or None, in_limbs, out_ref); refs address element `first`."""
        self.engine.prf_jobs_dev(it, n, n_jobs, [(a, m, first, count, self._a(i), il, self._a(o)) for a, m, first, count, i, il, o in jobs])

    # ---- packed reduce (the arbiter's one-big-integer add, jzf_aggregator.py:406-419) ----
    def pack(self, n, src, dst):
        self.engine.pack_dev(n, self._a(src), self._a(dst))

    def unpack(self, n, src, dst):
        self.engine.unpack_dev(n, self._a(src), self._a(dst))

    def aggregate_packed(self, srcs, n_limbs, total_bits, out):
        self.engine.aggregate_packed_dev([self._a(r) for r in srcs], n_limbs, total_bits, self._a(out))

    def packed_probe(self, x, n_limbs, info):
        self.engine.packed_probe_dev(n_limbs, self._a(x), self._a(info))

    def packed_resolve_carry(self, x, n_limbs, total_bits, infos, n_below, stride_words=3):
        """stride_words = -3: `infos` points at the LOWEST slice's triple and the more significant ones lie in front of it."""
        if stride_words == 3:
            self.engine.packed_resolve_carry_dev(n_limbs, total_bits, self._a(infos), n_below, self._a(x))
        else:
            self.engine.packed_resolve_carry_strided_dev(n_limbs, total_bits, self._a(infos), n_below, stride_words, self._a(x))

    # ---- the sparse round by position ranges ----
    def sparse_span(self):
        return self.engine.sparse_span()

    def sparse_bounds(self, total, locs, ks, handle=None):
        """The span bounds of the round's location lists (refs of uint32 lists): a handle, recomputed in place when one is passed."""
        ptrs = [self._a(r) for r in locs]
        keep = [r[0] for r in locs]                       # the buffers behind the addresses: the handle keeps them alive
        if handle is not None:
            return handle.recompute(ptrs, ks, keep=keep)
        return self.engine.span_bounds(total, ptrs, ks, keep=keep)

    def sparse_encrypt_aggregate(self, it, idx, locs, ks, pts, pt_limbs, zeros, total, n_jobs, cts, agg, bounds, first, count):
        self.engine.sparse_encrypt_aggregate_dev(it, idx, [self._a(r) for r in locs], ks, [self._a(r) for r in pts], pt_limbs, zeros, total, n_jobs,
                                                 [self._a(r) for r in cts], self._a(agg), bounds=bounds, position_range=(first, count))

    def sparse_decrypt(self, it, locs, ks, total, n_jobs, agg, out, bounds, first, count):
        self.engine.sparse_decrypt_dev(it, [self._a(r) for r in locs], ks, total, n_jobs, self._a(agg), self._a(out), bounds=bounds,
                                       position_range=(first, count))

    def zero(self, ref, words):
        self.engine._check(self.engine._lib.flashe_memset_dev(self.engine._h, self._a(ref), 0, int(words) * 8))

    # ---- exchange ----
    def all_to_all(self, send, send_stride, recv, recv_stride, words, side=False):
        """Piece p (`words` long, at send + p * send_stride) -> rank p; strides in words."""
        self.comm.all_to_all(self._eng(side), self._a(send), 8 * send_stride, self._a(recv), 8 * recv_stride, 8 * words)

    def all_gather(self, send, recv, words, side=False):
        self.comm.all_gather(self._eng(side), self._a(send), self._a(recv), 8 * words)

    def allreduce_modadd(self, ref, words, side=False):
        """int_bits <= 64: the vector at ref becomes the mod-2^b sum of all ranks' vectors, on every rank."""
        self.comm.allreduce_modadd(self._eng(side), self._a(ref), words)

    def allreduce(self, value, op=0):
        if self.comm is None:
            self.sync()
            return float(value)
        return self.comm.allreduce(self.engine, value, op)

    def barrier(self):
        if self.comm is None:
            self.sync()
        else:
            self.comm.barrier(self.engine)


def make_hip_ops(key, int_bits, local_rank=0, two_streams=True, with_comm=None):
    """Engines (+ RCCL communicator when WORLD_SIZE > 1 or with_comm) for one rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    eng = Engine(key, int_bits, device=local_rank)
    side = Engine(key, int_bits, device=local_rank) if two_streams else None
    comm = RcclComm.from_env(eng) if (world > 1 or with_comm) else None
    return HipOps(eng, side, comm)


# ------------------------------------------------------------------------------------------------------------------
# the round
# ------------------------------------------------------------------------------------------------------------------
class ShardedRound:
    """One FLASHE round (encrypt x clients -> aggregate -> decrypt) with clients sharded over ranks.

    `clients` = the global client numbers (= cipher idx) this rank hosts, `total_clients` = how many upload in all; the
    per-rank counts may differ (deal_clients).  An int means "this many per rank, rank r hosts r*k .. r*k + k - 1"."""

    def __init__(self, ops, n, int_bits, clients, n_jobs, rank=0, world=1, total_clients=None, scheme=SCHEME_DOUBLE,
                 force_collectives=False, collective="all_to_all", shard="clients"):
        if shard not in ("clients", "elements"):
            raise ValueError(f"unknown shard {shard!r}")
        self.shard = shard
        if shard == "elements":
            # rank g owns a contiguous element slice of EVERY client's vector and plays every client on it.  Mask
            # streams are position-indexed, so the element-wise aggregate needs no exchange at all (an optional all-gather hands the
            # decrypted slices to every rank) and the packed aggregate one carry triple per slice boundary.  `clients` = how many
            # clients upload (or their numbers 0 .. C - 1); every rank passes the same.
            total_clients = clients if isinstance(clients, int) else len(clients)
            clients = list(range(total_clients))
        elif isinstance(clients, int):
            clients, total_clients = list(range(rank * clients, (rank + 1) * clients)), world * clients
        self.ops, self.n, self.b, self.n_jobs = ops, n, int_bits, n_jobs
        self.clients, self.cpr = list(clients), len(clients)
        self.total = int(total_clients if total_clients is not None else len(clients))
        self.rank, self.world, self.scheme = rank, world, scheme
        # world == 1 normally skips the exchange; force_collectives runs it anyway (a 1-rank all-to-all /
        # all-gather), which lets a single-GPU box exercise the exact RCCL code path
        self.exchange = world > 1 or force_collectives
        self.L = L = limbs_of(int_bits)
        # how the partial aggregates meet (sequential schedule): "all_to_all" = reduce-scatter built from point-to-point transfers
        # + local mod-add + all-gather (any int_bits: every GPU pair has its own xGMI link); "allreduce" = RCCL's own
        # ncclAllReduce(uint64, sum) + mask, int_bits <= 64 only (SURVEY.md section 8e names both)
        if collective not in ("all_to_all", "allreduce"):
            raise ValueError(f"unknown collective {collective!r}")
        if collective == "allreduce" and L != 1:
            raise ValueError("collective='allreduce' needs int_bits <= 64 (RCCL has no 128-bit integer sum)")
        self.collective = collective
        self.slice = slice_len(n, world)
        self.padded = self.slice * world
        # the local ciphertexts are equally spaced in ONE allocation: the fused reduce + decrypt walks them by stride
        held = self.slice if shard == "elements" else n                  # elements of a client vector this rank holds
        self.ct_stride = stride = (held * L + 1) // 2 * 2                # every vector 16-byte aligned
        self.ct_all = ops.alloc(max(self.cpr, 1) * stride)
        self.ct = [(self.ct_all, c * stride) for c in range(self.cpr)]
        self.partial = ops.alloc((self.slice if shard == "elements" else self.padded) * L)   # local aggregate (client sharding: padded to world slices)
        self.recv = ops.alloc(self.padded * L) if (self.exchange and shard == "clients") else None
        self.agg_slice = ops.alloc(self.slice * L)
        self.dec_slice = ops.alloc(self.slice * L)
        self.result = ops.alloc(self.padded * L)                         # plaintext aggregate on every rank
        self.first = min(rank * self.slice, n)
        self.count = min(self.slice, n - self.first)

    def total_clients(self):
        return self.total

    def _prefixes(self):
        uploaded = list(range(self.total))
        return telescope(uploaded) if self.scheme == SCHEME_DOUBLE else ([], uploaded)

    @staticmethod
    def _at(ref, words):
        return (ref[0], ref[1] + words)

    # ---- sequential schedule -------------------------------------------------------------------------------------
    def encrypt_phase(self, it, pts, pt_limbs, partial_agg=False):
        """Every local client encrypts its vector (cipher idx = global client number).  partial_agg: the same launch also writes
        this rank's partial aggregate, the sum of its clients' ciphertexts, to self.partial ("each GPU encrypts
        and locally mod-adds its share"), so that reduce_decrypt_phase(partial_agg=True) does not re-read the ciphertexts."""
        if self.cpr:
            self.ops.encrypt_batch(it, self.clients, self.scheme, self.n, self.n_jobs, pts, pt_limbs, self.ct,
                                   sum_out=(self.partial, 0) if partial_agg else None)
        elif partial_agg:
            self.ops.zero((self.partial, 0), self.n * self.L)

    def _local_reduce(self, count, out, first=0, side=False):
        """out[first ..] = sum of the local ciphertexts on elements [first, first + count) (zeros on a rank without clients)."""
        L = self.L
        if self.cpr:
            self.ops.aggregate([self._at(r, first * L) for r in self.ct], count, self._at(out, first * L), side=side)
        else:
            self.ops.zero(self._at(out, first * L), count * L)

    def reduce_decrypt_phase(self, it, partial_agg=False):
        """Reduce (+ exchange) with the last reduce fused into the decrypt (one pass over its operands): without an exchange the
        C local ciphertexts, with one the W received pieces of the owned slice.  partial_agg: self.partial already holds the local
        sum (encrypt_phase(partial_agg=True)): it is decrypted, or exchanged, as it is."""
        add_idx, minus_idx = self._prefixes()
        ops, n, W, L = self.ops, self.n, self.world, self.L
        if not self.exchange:
            if partial_agg:
                ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, 0, n, (self.partial, 0), (self.result, 0))
            else:
                ops.aggregate_decrypt(it, add_idx, minus_idx, n, self.n_jobs, 0, n, self.ct, (self.partial, 0), (self.result, 0))
            return self.result
        if not partial_agg:
            self._local_reduce(n, (self.partial, 0))
        if self.collective == "allreduce":
            # every rank ends up with the whole aggregate and decrypts it itself: at int_bits <= 64 the decrypt is 2 / m AES blocks
            # per element, cheaper than gathering decrypted slices
            ops.allreduce_modadd((self.partial, 0), n)
            ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, 0, n, (self.partial, 0), (self.result, 0))
            return self.result
        ops.all_to_all((self.partial, 0), self.slice * L, (self.recv, 0), self.slice * L, self.slice * L)
        if self.count > 0:
            ops.aggregate_decrypt(it, add_idx, minus_idx, n, self.n_jobs, self.first, self.count,
                                  [(self.recv, g * self.slice * L) for g in range(W)], (self.agg_slice, 0), (self.dec_slice, 0))
        ops.all_gather((self.dec_slice, 0), (self.result, 0), self.slice * L)
        return self.result

    def run(self, it, pts, pt_limbs, partial_agg=False):
        """pts: refs of this rank's plaintext vectors (one per local client).  Returns the buffer holding the decrypted
        aggregate (first n*L words valid).  Two PRF launches per round on one GPU: every local encrypt, then reduce + decrypt
        (partial_agg: every local encrypt + their sum, then the decrypt of that sum)."""
        if self.shard == "elements":
            return self.run_elements(it, pts, pt_limbs, partial_agg=partial_agg)
        self.encrypt_phase(it, pts, pt_limbs, partial_agg)
        return self.reduce_decrypt_phase(it, partial_agg)

    # ---- element sharding ---------------------------------------------------------------------------
    def element_range(self, packed=False):
        """(first, count): the elements of every client vector this rank owns under shard="elements".  Element-wise aggregate:
        slices of `slice_len(n, world)` elements counted from the FRONT.  Packed aggregate: counted from the END -- rank g owns
        [max(0, n - (W - g) S), max(0, n - (W - 1 - g) S)) -- so that every slice boundary is a whole number of 64-bit limbs above bit 0
        of the packed integer (element n - 1 sits at bit 0, element 0 is the most significant, jzf_weights.py:59-62): S is a multiple
        of 256 elements, hence S * b a multiple of 64 bits whatever b is."""
        if not packed:
            return self.first, self.count
        n, W, S, g = self.n, self.world, self.slice, self.rank
        lo, hi = max(0, n - (W - g) * S), max(0, n - (W - 1 - g) * S)
        return lo, hi - lo

    def run_elements(self, it, pts, pt_limbs, partial_agg=True, gather=True):
        """The round with ELEMENTS sharded over the ranks: every rank runs the full client chain (C + 1 PRF streams, whatever the
        number of GPUs -- client sharding loses the stream sharing when clients are spread thin) on its own slice of the vectors,
        reduces and decrypts that slice, and nothing is exchanged for it: the one collective is the optional all-gather of the
        decrypted slices (gather=False leaves every rank with its slice in the returned buffer).  pts: refs of this rank's SLICE of
        every client's plaintext (element `element_range()[0]` at the ref), clients in order 0 .. C - 1.  partial_agg (default): the
        encrypt launch also writes the slice of the ciphertexts' sum, the second launch decrypts it."""
        assert self.shard == "elements", 'run_elements needs ShardedRound(shard="elements")'
        ops, n, L = self.ops, self.n, self.L
        first, count = self.first, self.count
        add_idx, minus_idx = self._prefixes()
        if count:
            ops.encrypt_batch_range(it, self.clients, self.scheme, n, self.n_jobs, first, count, pts, pt_limbs, self.ct,
                                    sum_out=(self.partial, 0) if partial_agg else None)
            if partial_agg:
                ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, first, count, (self.partial, 0), (self.dec_slice, 0))
            else:
                ops.aggregate_decrypt(it, add_idx, minus_idx, n, self.n_jobs, first, count, self.ct, (self.partial, 0), (self.dec_slice, 0))
        if not (gather and self.exchange):
            return self.dec_slice
        ops.all_gather((self.dec_slice, 0), (self.result, 0), self.slice * L)
        return self.result

    def _elements_packed_buffers(self):
        if getattr(self, "ek_nl", None) is not None:
            return
        ops, W, S, L = self.ops, self.world, self.slice, self.L
        lo, cnt = self.element_range(packed=True)
        self.ek_lo, self.ek_cnt = lo, cnt
        self.ek_top = lo == 0                                     # the most significant non-empty slice: its carry-out is dropped
        self.ek_nl = nl = (cnt * self.b + 63) // 64
        even = (nl + 2 + 1) // 2 * 2
        self.ek_packed = [ops.alloc(even) for _ in range(self.cpr)]     # every packed slice with two zero limbs on top
        self.ek_sum = ops.alloc(even)
        self.ek_info = ops.alloc(4)
        self.ek_infos = ops.alloc(3 * W + 1)
        self.ek_agg = ops.alloc(max(cnt, 1) * L)
        self.ek_piece = ops.alloc(S * L)                          # the decrypted slice, right-aligned in an S-element piece
        self.ek_all = ops.alloc(W * S * L) if self.exchange else None

    def run_elements_packed(self, it, pts, pt_limbs, gather=True):
        """The element-sharded round with the arbiter's PACKED reduce (jzf_aggregator.py:406-419: every model one n*b-bit integer,
        carries cross element boundaries).  Each rank packs and adds its slice of the C ciphertexts one limb wider than the slice
        (the extra limb is the slice's carry-out); the ONLY exchange for the reduce is an all-gather of one (low limb, all-ones flag,
        carry-out) triple per rank, from which every rank resolves its carry-in on the device -- the slices less significant than
        rank g's are those of ranks g + 1 .. W - 1 (element 0 is the most significant), hence the backward walk over the gathered
        triples.  pts: refs of this rank's slice `element_range(packed=True)` of every client's plaintext.  Returns a buffer whose
        first n*L words are the decrypted aggregate (gather=True), or this rank's slice right-aligned in an S-element piece."""
        assert self.shard == "elements", 'run_elements_packed needs ShardedRound(shard="elements")'
        ops, n, L, W, S, b = self.ops, self.n, self.L, self.world, self.slice, self.b
        self._elements_packed_buffers()
        lo, cnt, nl, top = self.ek_lo, self.ek_cnt, self.ek_nl, self.ek_top
        add_idx, minus_idx = self._prefixes()
        ops.zero((self.ek_info, 0), 3)
        if cnt:
            ops.encrypt_batch_range(it, self.clients, self.scheme, n, self.n_jobs, lo, cnt, pts, pt_limbs, self.ct)
            for c in range(self.cpr):
                ops.pack(cnt, self.ct[c], (self.ek_packed[c], 0))
            rows = [(p, 0) for p in self.ek_packed]
            if top:
                ops.aggregate_packed(rows, nl, cnt * b, (self.ek_sum, 0))
            else:
                ops.aggregate_packed(rows, nl + 1, 64 * (nl + 1), (self.ek_sum, 0))
                ops.packed_probe((self.ek_sum, 0), nl + 1, (self.ek_info, 0))
        if self.exchange:
            ops.all_gather((self.ek_info, 0), (self.ek_infos, 0), 3)
            below = W - 1 - self.rank
            if cnt and below:
                ops.packed_resolve_carry((self.ek_sum, 0), nl, cnt * b if top else 64 * nl, (self.ek_infos, 3 * (W - 1)), below, stride_words=-3)
        if cnt:
            ops.unpack(cnt, (self.ek_sum, 0), (self.ek_agg, 0))
            ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, lo, cnt, (self.ek_agg, 0), (self.ek_piece, (S - cnt) * L))
        if not (gather and self.exchange):
            return ops.view(self.ek_piece, (S - cnt) * L)
        ops.all_gather((self.ek_piece, 0), (self.ek_all, 0), S * L)
        return ops.view(self.ek_all, (W * S - n) * L)

    # ---- chunk-pipelined schedules -------------------------------------------------------------------------------
    def _pipe_buffers(self, chunks):
        """Block-cyclic ownership for the pipelined schedules: the vector is cut into `chunks` contiguous chunks of
        world * sub elements and rank g owns piece g of EVERY chunk, so each chunk's all-to-all / all-gather works on
        one contiguous block."""
        key = ("pipe", chunks)
        if getattr(self, "_pipe_key", None) == key:
            return
        # the buffers of every chunk count used so far are kept: a calibration that alternates between chunk counts must not pay
        # (or time) a gigabyte of allocations per switch
        cache = self.__dict__.setdefault("_pipe_cache", {})
        if getattr(self, "_pipe_key", None) is not None:
            cache[self._pipe_key] = {k: getattr(self, k) for k in ("p_sub", "p_chunk", "p_partial", "p_result", "p_recv", "p_agg", "p_dec", "p_dmask")}
        if key in cache:
            for k, v in cache[key].items():
                setattr(self, k, v)
            self._pipe_key = key
            return
        n, W, L, ops = self.n, self.world, self.L, self.ops
        sub = ((n + chunks * W - 1) // (chunks * W) + ALIGN - 1) // ALIGN * ALIGN
        self.p_sub, self.p_chunk = sub, sub * W
        padded = self.p_chunk * chunks
        self.p_partial = ops.alloc(padded * L)
        self.p_result = ops.alloc(padded * L)
        self.p_recv = ops.alloc(padded * L) if self.exchange else None
        self.p_agg = ops.alloc(chunks * sub * L) if self.exchange else None
        self.p_dec = ops.alloc(chunks * sub * L) if self.exchange else None
        self.p_dmask = None
        self._pipe_key = key

    def run_pipelined(self, it, pts, pt_limbs, chunks=4, batch_events=None):
        """The round with everything after the last client's encrypt hidden under it, chunk by chunk: the last client
        encrypts chunk q on the main stream; on the side stream chunk q is reduced locally, reduce-scattered (all-to-all +
        mod-add of the received pieces), the owned piece is decrypted and all-gathered -- while the main stream already
        encrypts chunk q + 1.  Same arithmetic as run(); only the schedule and the (block-cyclic) ownership differ."""
        ops, n, L, W = self.ops, self.n, self.L, self.world
        self._pipe_buffers(chunks)
        sub, chunk = self.p_sub, self.p_chunk
        last = self.cpr - 1
        if last > 0:
            if batch_events:                       # (start, stop) engine events bracketing the batched launch
                ops.engine.record(batch_events[0])
            ops.encrypt_batch(it, self.clients[:last], self.scheme, n, self.n_jobs, pts[:last], pt_limbs, self.ct[:last])
            if batch_events:
                ops.engine.record(batch_events[1])
        add_idx, minus_idx = self._prefixes()
        for q in range(chunks):
            first = q * chunk
            cnt = max(0, min(chunk, n - first))
            if cnt and self.cpr:
                ops.encrypt_range(it, self.clients[last], self.scheme, n, self.n_jobs, first, cnt,
                                  self._at(pts[last], first * pt_limbs), pt_limbs, self._at(self.ct[last], first * L))
            ops.signal(f"enc{q}")
            ops.wait(f"enc{q}", side=True)
            if cnt:
                self._local_reduce(cnt, (self.p_partial, 0), first=first, side=True)
            if self.exchange:
                blk = first * L
                ops.all_to_all((self.p_partial, blk), sub * L, (self.p_recv, blk), sub * L, sub * L, side=True)
                ops.aggregate([(self.p_recv, blk + g * sub * L) for g in range(W)], sub, (self.p_agg, q * sub * L), side=True)
                gfirst = first + self.rank * sub
                gcnt = max(0, min(sub, n - gfirst))
                if gcnt:
                    ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, gfirst, gcnt, (self.p_agg, q * sub * L),
                                      (self.p_dec, q * sub * L), side=True)
                ops.all_gather((self.p_dec, q * sub * L), (self.p_result, blk), sub * L, side=True)
            ops.signal(f"done{q}", side=True)
        for q in range(chunks):
            ops.wait(f"done{q}")
            if not self.exchange:
                # single rank: the decrypt is as heavy as one encrypt, so it stays on the main stream (right
                # behind the encrypts) and only the HBM-bound reduce is hidden on the side stream
                first = q * chunk
                cnt = max(0, min(chunk, n - first))
                if cnt:
                    ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, first, cnt, (self.p_partial, first * L),
                                      (self.p_result, first * L))
        return self.p_result

    def run_fused(self, it, pts, pt_limbs, chunks=4, launch_events=None):
        """Double mask, nobody dropped.  Per chunk, ONE launch on the main stream does all the mask arithmetic: the local
        clients' encrypts of the chunk (one chain: consecutive clients share their streams) plus D = term(iter, C) -
        term(iter, 0) on the piece of the chunk this rank publishes -- the reference's prepare_decrypt
        (jzf_flashe.py:633-666), kept as the fused difference.  The side stream then only runs HBM-bound adds and the
        exchange: the reduce takes D as one more operand, so what it writes IS the plaintext aggregate (decrypt's
        `value + add - minus`, :570-571).  Chunk q's reduce / exchange hides under chunk q + 1's launch.  Same results
        as run().  launch_events: optional list of `chunks` entries, each None or a (start, stop) engine event pair
        that brackets that chunk's launch."""
        assert self.scheme == SCHEME_DOUBLE, "run_fused needs the double mask (one add / one minus prefix)"
        ops, n, L, W = self.ops, self.n, self.L, self.world
        self._pipe_buffers(chunks)
        sub, chunk = self.p_sub, self.p_chunk
        if self.p_dmask is None:
            self.p_dmask = ops.alloc(chunks * sub * L if self.exchange else chunks * chunk * L)
        C = self.total
        for q in range(chunks):
            first = q * chunk
            cnt = max(0, min(chunk, n - first))
            # (a chunk may lie entirely beyond the end of the vector: the chunks are padded to world * sub elements)
            jobs = [(cl, cl + 1, first, cnt, self._at(pts[c], first * pt_limbs), pt_limbs, self._at(self.ct[c], first * L))
                    for c, cl in enumerate(self.clients)] if cnt else []
            if self.exchange:
                gfirst = min(first + self.rank * sub, n)
                jobs.append((C, 0, gfirst, max(0, min(sub, n - gfirst)), None, 0, (self.p_dmask, q * sub * L)))
            elif cnt:
                jobs.append((C, 0, first, cnt, None, 0, (self.p_dmask, first * L)))
            ev = launch_events[q] if launch_events else None
            if ev:
                ops.engine.record(ev[0])
            ops.prf_jobs(it, n, self.n_jobs, jobs)
            if ev:
                ops.engine.record(ev[1])
            ops.signal(f"enc{q}")
            ops.wait(f"enc{q}", side=True)
            if not self.exchange:
                if cnt:
                    ops.aggregate([self._at(r, first * L) for r in self.ct] + [(self.p_dmask, first * L)], cnt,
                                  (self.p_result, first * L), side=True)
            else:
                if cnt:
                    self._local_reduce(cnt, (self.p_partial, 0), first=first, side=True)
                blk = first * L
                ops.all_to_all((self.p_partial, blk), sub * L, (self.p_recv, blk), sub * L, sub * L, side=True)
                ops.aggregate([(self.p_recv, blk + g * sub * L) for g in range(W)] + [(self.p_dmask, q * sub * L)], sub,
                              (self.p_dec, q * sub * L), side=True)
                ops.all_gather((self.p_dec, q * sub * L), (self.p_result, blk), sub * L, side=True)
            ops.signal(f"done{q}", side=True)
        for q in range(chunks):
            ops.wait(f"done{q}")
        return self.p_result

    # ---- packed reduce -------------------------------------------------------------------------------------------
    def _packed_buffers(self):
        if getattr(self, "k_sl", None) is not None:
            return
        n, W, ops = self.n, self.world, self.ops
        self.k_bits = n * self.b
        self.k_nl = nl = (self.k_bits + 63) // 64
        # limb slices of the packed integer: rank g owns limbs [g * sl, (g + 1) * sl) (even count: 16-byte accesses)
        self.k_sl = sl = ((nl + W - 1) // W + 1) // 2 * 2
        self.k_packed = [ops.alloc(nl + (nl & 1)) for _ in range(self.cpr)]
        self.k_partial = ops.alloc(W * sl)                 # this rank's packed sum; limbs beyond nl stay zero
        self.k_full = ops.alloc(W * sl)                    # the packed aggregate on every rank
        self.k_agg = ops.alloc(n * self.L)
        if self.exchange:
            self.k_rows = ops.alloc(W * (sl + 2))          # received slices, each with two zero limbs on top
            self.k_sum = ops.alloc(sl + 2)
            self.k_info = ops.alloc(4)
            self.k_infos = ops.alloc(3 * W + 1)

    def run_packed(self, it, pts, pt_limbs):
        """The round as a dense FLASHE job runs it: every client model travels as ONE n*b-bit integer and the arbiter adds
        those integers mod 2^(n*b) (jzf_aggregator.py:406-419), so carries cross element boundaries.  Across ranks
        (packed variant): local packed sum; all-to-all of limb slices (received straight into rows with
        two spare limbs on top); rank g adds its W slices one limb wider than the slice, which leaves the slice's
        carry-out in the extra limb; an all-gather of (low limb, all-ones flag, carry-out) per slice lets every rank
        derive its carry-in ON THE DEVICE, including the case where a carry ripples through a whole slice
        (flashe_packed_resolve_carry_dev); all-gather of the slices.  Every rank then unpacks and decrypts the
        aggregate, as every client of the reference does."""
        if self.shard == "elements":
            return self.run_elements_packed(it, pts, pt_limbs)
        ops, n, W = self.ops, self.n, self.world
        self._packed_buffers()
        nl, sl, bits = self.k_nl, self.k_sl, self.k_bits
        self.encrypt_phase(it, pts, pt_limbs)
        for c in range(self.cpr):
            ops.pack(n, self.ct[c], (self.k_packed[c], 0))
        if self.cpr:
            ops.aggregate_packed([(p, 0) for p in self.k_packed], nl, bits, (self.k_partial, 0))
        else:
            ops.zero((self.k_partial, 0), W * sl)
        total = self.k_partial
        if self.exchange:
            ops.all_to_all((self.k_partial, 0), sl, (self.k_rows, 0), sl + 2, sl)
            lo = self.rank * sl
            cnt = max(0, min(sl, nl - lo))
            last = lo + sl >= nl                         # the top slice: its carry-out is dropped (mod 2^(n*b))
            rows = [(self.k_rows, g * (sl + 2)) for g in range(W)]
            ops.zero((self.k_info, 0), 3)
            if cnt and last:
                ops.aggregate_packed(rows, cnt, bits - 64 * lo, (self.k_sum, 0))
            elif cnt:
                ops.aggregate_packed(rows, sl + 1, 64 * (sl + 1), (self.k_sum, 0))
                ops.packed_probe((self.k_sum, 0), sl + 1, (self.k_info, 0))
            ops.all_gather((self.k_info, 0), (self.k_infos, 0), 3)
            if cnt and self.rank:                        # slices below this one are full, never the top slice
                ops.packed_resolve_carry((self.k_sum, 0), cnt, bits - 64 * lo if last else 64 * sl, (self.k_infos, 0), self.rank)
            ops.all_gather((self.k_sum, 0), (self.k_full, 0), sl)
            total = self.k_full
        ops.unpack(n, (total, 0), (self.k_agg, 0))
        add_idx, minus_idx = self._prefixes()
        ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, 0, n, (self.k_agg, 0), (self.result, 0))
        return self.result


class SparseShardedRound:
    """BASELINE config 5's round -- top-s % uploads, single mask over compact positions -- sharded by POSITION ranges of the dense vector
    (SURVEY.md 8e (i) applied to the sparse path): rank g owns the positions [g S, (g + 1) S), S a whole number of spans, and plays EVERY
    client on them: the entries of each (sorted) location list that fall into its range are encrypted and summed in one pass
    (flashe_sparse_encrypt_aggregate_range_dev), the range of the dense minus-mask is rebuilt and subtracted in another
    (flashe_sparse_decrypt_range_dev).  PRF counters are compact positions of the WHOLE list, so a rank's ciphertext entries are exactly
    those a single GPU would have produced; nothing is exchanged for the aggregate, the one collective is the optional all-gather of the
    decrypted ranges.  Every rank holds all location lists (they are what the arbiter redistributes for the decrypt anyway) and the
    plaintext values of the entries it owns."""

    def __init__(self, ops, total, int_bits, n_clients, n_jobs, rank=0, world=1):
        self.ops, self.total, self.b, self.C, self.n_jobs, self.rank, self.world = ops, int(total), int_bits, int(n_clients), n_jobs, rank, world
        if int_bits <= 64:
            raise ValueError("the position-sharded sparse round runs at int_bits > 64 (the passes with the PRF inside the span reduce)")
        self.L = 2
        span = ops.sparse_span()
        n_spans = -(-self.total // span)
        self.slice = span * (-(-n_spans // world))
        self.first = min(self.total, rank * self.slice)
        self.count = min(self.total, (rank + 1) * self.slice) - self.first
        self.agg = ops.alloc(max(self.slice, 1) * self.L)
        self.dec = ops.alloc(max(self.slice, 1) * self.L)
        self.result = ops.alloc(world * self.slice * self.L) if world > 1 else None
        self.bounds = None

    def position_range(self):
        return self.first, self.count

    def run(self, it, locs, ks, pts, pt_limbs, zeros, cts, idx=None, gather=True):
        """locs / pts / cts: refs per client (uint32 list, compact plaintexts, compact ciphertexts -- whole vectors; only the owned entries
        of pts are read and of cts written); zeros: the plain quantised zero of every upload.  Returns the ref of the decrypted range
        (gather=False / one rank) or of the whole decrypted vector."""
        ops = self.ops
        idx = list(range(self.C)) if idx is None else idx
        self.bounds = ops.sparse_bounds(self.total, locs, ks, self.bounds)
        if self.count:
            ops.sparse_encrypt_aggregate(it, idx, locs, ks, pts, pt_limbs, zeros, self.total, self.n_jobs, cts, (self.agg, 0), self.bounds,
                                         self.first, self.count)
            ops.sparse_decrypt(it, locs, ks, self.total, self.n_jobs, (self.agg, 0), (self.dec, 0), self.bounds, self.first, self.count)
        if not (gather and self.world > 1):
            return self.dec
        ops.all_gather((self.dec, 0), (self.result, 0), self.slice * self.L)
        return self.result
