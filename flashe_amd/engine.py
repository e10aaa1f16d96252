"""Thin object layer over the C ABI: one `Engine` = one flashe_ctx (device + stream + key +
int_bits); `DeviceBuffer` = a vector resident in HBM.  No arithmetic happens in Python.
"""
import ctypes
import os
import threading
import weakref

import numpy as np

from . import _lib
from ._lib import SCHEME_DOUBLE, SCHEME_SINGLE, FlasheError, PrfJob, c_int, c_u32, c_u32p, c_u64, c_u64p, c_vp

__all__ = ["Engine", "DeviceBuffer", "DeviceVector", "limbs_of", "SCHEME_SINGLE", "SCHEME_DOUBLE", "FlasheError",
           "chunks", "telescope", "prp_block"]


def limbs_of(int_bits):
    return 2 if int_bits > 64 else 1


def _u32_list(vals):
    arr = (c_u32 * max(len(vals), 1))(*[int(v) & 0xFFFFFFFF for v in vals])
    return ctypes.cast(arr, c_u32p), arr


def chunks(n, n_jobs):
    """chunks_idx(range(n), n_jobs) boundaries -- jzf_flashe.py:12-16."""
    out = (c_u64 * (n_jobs + 1))()
    rc = _lib.load().flashe_chunks(n, n_jobs, ctypes.cast(out, c_u64p))
    if rc:
        raise FlasheError(rc, "flashe_chunks: bad arguments")
    return [int(v) for v in out]


def telescope(raw_idx_list):
    """(add_idx, minus_idx) of set_idx_list(mode='decrypt') -- jzf_flashe.py:356-367."""
    k = len(raw_idx_list)
    raw, _r = _u32_list(list(raw_idx_list))
    add = (c_u32 * max(k, 1))()
    minus = (c_u32 * max(k, 1))()
    runs = c_int(0)
    rc = _lib.load().flashe_telescope(raw, k, ctypes.cast(add, c_u32p), ctypes.cast(minus, c_u32p), ctypes.byref(runs))
    if rc:
        raise FlasheError(rc, "flashe_telescope: bad arguments")
    return [int(v) for v in add[:runs.value]], [int(v) for v in minus[:runs.value]]


def prp_block(key, block):
    """AES-256-ECB of one 16-byte block on the host -- jzf_aes_prp.py:24-30."""
    out = (ctypes.c_uint8 * 16)()
    rc = _lib.load().flashe_prp_block((ctypes.c_uint8 * 32).from_buffer_copy(bytes(key)),
                                      (ctypes.c_uint8 * 16).from_buffer_copy(bytes(block)), out)
    if rc:
        raise FlasheError(rc, "flashe_prp_block: bad arguments")
    return bytes(out)


class _HostPool:
    """Recycled host memory behind the result arrays of the host-array API.  A fresh 160 MB array costs 4-7 ms of page faults on its
    first transfer -- more than the 2.9 ms PCIe Gen5 needs to fill it (tests/perf/e2e_calls.py) -- and NumPy hands large arrays
    straight back to the OS, so every call would pay that again.  Blocks the caller has dropped (all views of the array dead) are
    kept and reused, up to FLASHE_HOST_POOL_MB (default 4096; 0 = plain np.empty).

    Page-locked blocks: the host-pointer twins pipeline a large call (upload + kernel of one chunk beside the download of the
    previous one) only into a PINNED result array -- 85 -> 70 ms for the config-2 round (tests/perf/e2e_pinned.py) -- but pinning
    160 MB costs 30 ms, which only a caller that comes back earns.  FLASHE_HOST_POOL_PINNED = auto (default): the first
    PIN_AFTER leases of a size class are pageable, from then on the class is served from page-locked blocks (a long-running job ends up
    all pinned, a one-shot script never pays); 1: always; 0: never.

    A recycled block is handed out UNINITIALISED, like np.empty: it still holds whatever result it carried before (a ciphertext, or a
    decrypted aggregate) until the call that leased it has overwritten all of it -- every Engine method does.  FLASHE_HOST_POOL_WIPE=1
    zeroes blocks as they return to the pool."""
    MIN_BYTES = 1 << 20
    STEP = 2 << 20
    PIN_AFTER = 16                  # > the 12 result arrays of one config-2 round: a single round never pays for pinning

    def __init__(self):
        # re-entrant: _release runs from weakref.finalize callbacks, and a garbage collection triggered INSIDE the locked region
        # (it allocates) may finalise another pooled array on the same thread
        self._lock = threading.RLock()
        self._free = {}                 # capacity -> [block]
        self._leases = {}               # capacity -> how many arrays of that class have been handed out
        self._held = 0                  # bytes parked in the free lists
        self._budget = max(0, int(os.environ.get("FLASHE_HOST_POOL_MB", "4096"))) << 20
        mode = os.environ.get("FLASHE_HOST_POOL_PINNED", "auto").strip().lower()
        self._pinned = "never" if mode in ("", "0", "off", "no") else "always" if mode in ("1", "on", "yes") else "auto"
        self._wipe = os.environ.get("FLASHE_HOST_POOL_WIPE", "0") not in ("", "0")

    class _Pinned:
        pinned = True

        def __init__(self, cap):
            p = c_vp()
            rc = _lib.load().flashe_host_alloc(cap, ctypes.byref(p))
            if rc:
                raise MemoryError("flashe_host_alloc")
            self.addr = p.value

        def __del__(self):
            try:
                _lib.load().flashe_host_free(self.addr)
            except Exception:
                pass

    class _Pageable:
        pinned = False

        def __init__(self, cap):
            self.arr = np.empty(cap, dtype=np.uint8)
            self.addr = self.arr.ctypes.data

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.MIN_BYTES or self._budget == 0:
            return np.zeros(shape, dtype=dtype)
        cap = (nbytes + self.STEP - 1) // self.STEP * self.STEP
        block = None
        with self._lock:
            count = self._leases[cap] = self._leases.get(cap, 0) + 1
            want_pinned = self._pinned == "always" or (self._pinned == "auto" and count > self.PIN_AFTER)
            lst = self._free.get(cap)
            if lst:
                # a page-locked block if there is one; a pageable one only while the class is not being pinned
                pick = next((i for i in range(len(lst) - 1, -1, -1) if lst[i].pinned), None)
                if pick is None and not want_pinned:
                    pick = len(lst) - 1
                if pick is not None:
                    block = lst.pop(pick)
                    self._held -= cap
                elif self._held + cap > self._budget:
                    self._held -= cap                         # make room for the pinned replacement: one parked pageable block goes
                    lst.pop()
        if block is None:
            try:
                block = (self._Pinned if want_pinned else self._Pageable)(cap)
            except MemoryError:
                try:
                    block = self._Pageable(cap)
                except MemoryError:
                    return np.empty(shape, dtype=dtype)
        lease = (ctypes.c_char * nbytes).from_address(block.addr)      # dies with the last view of the array
        fin = weakref.finalize(lease, self._release, block, cap)
        fin.atexit = False
        return np.frombuffer(lease, dtype=dtype).reshape(shape)

    def _release(self, block, cap):
        if self._wipe:
            ctypes.memset(block.addr, 0, cap)
        with self._lock:
            lst = self._free.get(cap)
            if lst is None:
                lst = self._free[cap] = []
            if self._held + cap <= self._budget:
                lst.append(block)
                self._held += cap
            elif block.pinned:
                # a page-locked block is worth more than a parked pageable one of its class
                for i, other in enumerate(lst):
                    if not other.pinned:
                        lst[i] = block
                        break
        # otherwise the block dies here

    def trim(self):
        """Give every parked block back."""
        with self._lock:
            self._free.clear()
            self._held = 0


_HOST_POOL = _HostPool()


def host_empty(shape, dtype=np.uint64):
    """An uninitialised result array, from the recycling pool when it is large (see _HostPool)."""
    return _HOST_POOL.empty(shape, dtype)


class DeviceBuffer:
    """`nbytes` of HBM owned by an Engine.  `.ptr` is the raw device address."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = c_vp()
        engine._check(engine._lib.flashe_dev_alloc(engine._h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr is not None and self.engine._h is not None:
            self.engine._check(self.engine._lib.flashe_dev_free(self.engine._h, self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.engine._check(self.engine._lib.flashe_memcpy_h2d(self.engine._h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def upload_at(self, offset, arr):
        """Copy `arr` to byte `offset` of the block (layers of a model going into one flat buffer)."""
        arr = np.ascontiguousarray(arr)
        assert 0 <= offset and offset + arr.nbytes <= self.nbytes
        if arr.nbytes:
            self.engine._check(self.engine._lib.flashe_memcpy_h2d(self.engine._h, self.ptr + int(offset), arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype=np.uint64, count=None):
        n = self.nbytes // np.dtype(dtype).itemsize if count is None else count
        out = host_empty(n, dtype)
        self.engine._check(self.engine._lib.flashe_memcpy_d2h(self.engine._h, out.ctypes.data, self.ptr, out.nbytes))
        return out


class DeviceVector:
    """A vector of `n` elements (L = `limbs` uint64 limbs each) that STAYS in HBM between calls of the drop-in API: what
    FlasheCipher.encrypt / aggregate / decrypt return with device=True and accept in place of an ndarray, so that a round moves
    plaintexts up once and the result down once instead of bouncing every ciphertext device -> host -> device
    (jzf_weights.py:334-338 -> jzf_flashe_block.py:142-174 is the call chain this shortens).  `.to_host()` downloads it.

    The producing engine records an event behind its last write; a consumer on another engine (another FlasheCipher = another
    stream) makes its stream wait for it on the device, the host never blocks."""

    def __init__(self, engine, n, limbs=None, buf=None, elem_bytes=8):
        """elem_bytes = 4: the COMPACT layout of int_bits <= 32 -- the same values as a uint32 array (flashe_*_u32_dev), half the bytes
        of the one-limb layout in HBM and over PCIe."""
        self.engine, self.n = engine, int(n)
        self.elem_bytes = int(elem_bytes)
        self.limbs = 1 if self.elem_bytes == 4 else int(limbs or engine.limbs)
        if buf is not None:
            self.buf = buf
        else:
            self.buf = engine.alloc(max(4 * self.n, 16)) if self.elem_bytes == 4 else engine.alloc_vec(self.n, self.limbs)
        self._ready = None

    @property
    def compact(self):
        return self.elem_bytes == 4

    def __len__(self):
        return self.n

    @property
    def ptr(self):
        return self.buf.ptr

    @property
    def shape(self):
        return (self.n, self.limbs)

    @property
    def device(self):
        return getattr(self.engine, "device", 0)

    @classmethod
    def from_host(cls, engine, arr):
        """Upload a uint64 array of shape [n] or [n, k] -- or a 1-D uint32 array (compact layout, int_bits <= 32)."""
        if isinstance(arr, np.ndarray) and arr.dtype == np.uint32:
            arr = np.ascontiguousarray(arr).reshape(-1)
            return cls(engine, arr.shape[0], 1, buf=engine.upload(arr), elem_bytes=4)
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        return cls(engine, arr.shape[0], arr.shape[1], buf=engine.upload(arr))

    def mark_ready(self):
        """Called by the producer after enqueueing the kernels that write the vector."""
        if hasattr(self.engine, "event"):
            if self._ready is None:
                self._ready = self.engine.event()
            self.engine.record(self._ready)
        return self

    def wait_on(self, engine):
        """Make `engine`'s stream wait (on the device) until the vector is complete."""
        if engine is not self.engine and self._ready is not None:
            engine.wait_event(self._ready)

    def to_host(self):
        """uint64 array [n, limbs] -- uint32 [n] for a compact vector -- (from the recycling host pool); blocks until the vector is
        complete."""
        if self.elem_bytes == 4:
            return self.buf.download(np.uint32, self.n)
        return self.buf.download(np.uint64, self.n * self.limbs).reshape(self.n, self.limbs)

    def widened(self, engine=None):
        """A compact vector as a one-limb DeviceVector (a device pass, flashe_widen_u32_dev); a one-limb vector as it is."""
        if self.elem_bytes != 4:
            return self
        eng = engine or self.engine
        self.wait_on(eng)
        out = DeviceVector(eng, self.n, 1)
        eng.widen_u32_dev(self.n, self.buf, out.buf)
        return out.mark_ready()

    def narrowed(self, engine=None):
        """A one-limb vector in the compact layout (flashe_narrow_u32_dev: the low 32 bits of every element)."""
        if self.elem_bytes == 4:
            return self
        assert self.limbs == 1
        eng = engine or self.engine
        self.wait_on(eng)
        out = DeviceVector(eng, self.n, 1, elem_bytes=4)
        eng.narrow_u32_dev(self.n, self.buf, out.buf)
        return out.mark_ready()

    # ---- the arbiter's literal reduce: reduce(lambda x, y: (x + y) % mod, models) (jzf_aggregator.py:419, :430) on weights objects whose
    # layers are handles (JZFOrderDictWeights.__add__ / __mod__, jzf_weights.py:340-357, :460-472 map `+` and `%` over the layers) ----
    def __add__(self, other):
        """Element-wise sum mod 2^int_bits of two device-resident vectors, as a new DeviceVector (one HBM-bound launch)."""
        if isinstance(other, (int, np.integer)) and int(other) == 0:
            return self                                            # sum(handles) starts from 0
        if not isinstance(other, DeviceVector):
            return NotImplemented
        eng = self.engine
        if len(other) != self.n:
            raise ValueError(f"operands could not be broadcast together with shapes ({self.n},) ({len(other)},) ")
        if other.device != self.device:
            raise ValueError(f"DeviceVector lives on device {other.device}, this one on device {self.device}")
        other.wait_on(eng)
        if self.compact and other.compact:
            out = DeviceVector(eng, self.n, 1, elem_bytes=4)
            eng.aggregate_elem_u32_dev([self.buf, other.buf], self.n, out.buf)
            return out.mark_ready()
        a, b = self.widened(eng), other.widened(eng)
        if a.limbs != eng.limbs or b.limbs != eng.limbs:
            raise ValueError(f"expected {eng.limbs} limbs per element, got {a.limbs} and {b.limbs}")
        out = DeviceVector(eng, self.n)
        eng.aggregate_elem_dev([a.buf, b.buf], self.n, out.buf)
        return out.mark_ready()

    __radd__ = __add__

    def __mod__(self, modulus):
        """`% (1 << int_bits)`: the vector itself -- every kernel already reduces mod 2^int_bits.  Any other modulus is refused."""
        if int(modulus) == 1 << self.engine.int_bits:
            return self
        raise ValueError(f"a DeviceVector of a {self.engine.int_bits}-bit cipher can only be reduced mod 2^{self.engine.int_bits}")

    def __del__(self):
        try:
            if self._ready is not None and self.engine._h is not None:
                self.engine.event_destroy(self._ready)
            self._ready = None
        except Exception:
            pass


class Graph:
    """A captured sequence of device calls (flashe_graph); launch() replays it on the engine's stream."""

    def __init__(self, engine, handle):
        self.engine, self._g = engine, handle

    def launch(self, iter_shift=0):
        """Replay.  iter_shift = r runs the captured calls as if each had been made with iter + r (round r after the
        captured one); 0 repeats the captured round, mask streams included."""
        self.engine._check(self.engine._lib.flashe_graph_launch_shifted(self.engine._h, self._g, int(iter_shift) & 0xFFFFFFFF))

    def __del__(self):
        try:
            if self._g:
                self.engine._lib.flashe_graph_destroy(self._g)
                self._g = None
        except Exception:
            pass


class PtrTable(object):
    """A table of device pointers built ONCE (the ctypes array a *_dev call hands to the library): what a per-round call takes in
    place of a list of buffers when the same clients' vectors are passed round after round -- for 50 clients building the arrays costs
    as much host time as the launches they describe.  Engine.ptr_table(items); keeps the buffers alive."""

    def __init__(self, engine, items):
        self.items = list(items)
        self.arr = (c_vp * max(len(self.items), 1))(*[engine._ptr(x) for x in self.items])
        self.ptr = ctypes.cast(self.arr, ctypes.POINTER(c_vp))

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)

    def __getitem__(self, i):
        return self.items[i]


class JobTable(object):
    """The job list of Engine.prf_jobs_dev built ONCE (Engine.job_table(jobs)): a hundred clients' jobs cost more host time to
    marshal than their launch takes on the device.  Keeps the buffers alive."""

    def __init__(self, engine, jobs):
        self.jobs = list(jobs)
        self.arr = (PrfJob * max(len(self.jobs), 1))()
        for e, job in enumerate(self.jobs):
            a, m, first, count, inp, in_limbs, out = job[:7]
            # optional tail: (n_in, in_stride_elements, sum_out_ptr or None) = the reduce of n_in vectors fused in
            n_in, in_stride, sum_out = job[7:10] if len(job) > 7 else (0, 0, None)
            self.arr[e] = PrfJob(a, 0 if m is None else m, 0 if m is None else 1, in_limbs, first, count,
                                 None if inp is None else engine._ptr(inp), engine._ptr(out), n_in, 0, in_stride,
                                 None if sum_out is None else engine._ptr(sum_out))

    def __len__(self):
        return len(self.jobs)


class U64Table(object):
    """The same for a per-client list of lengths (Engine.u64_table(values))."""

    def __init__(self, values):
        self.values = [int(v) for v in values]
        self.arr = (c_u64 * max(len(self.values), 1))(*self.values)

    def __len__(self):
        return len(self.values)

    def __iter__(self):
        return iter(self.values)

    def __getitem__(self, i):
        return self.values[i]


def _u64_array(vals):
    return vals.arr if isinstance(vals, U64Table) else (c_u64 * max(len(vals), 1))(*[int(v) for v in vals])


class SpanBounds:
    """flashe_span_bounds: where every client's strictly increasing location list enters every span of the dense vector, computed once
    per round's lists.  The handle is a table ABOUT the lists' contents: the C side can check pointers, lengths and total, not that the
    entries are still the ones the table was built from -- after ANY in-place change of a list call recompute() before the next sparse
    call that takes the handle (SparseShardedRound does, every round).  The handle keeps the list buffers alive (`keep`: the buffer
    objects when `locs` are bare addresses)."""

    def __init__(self, engine, total, locs, ks, keep=None):
        self.engine = engine
        self._keep = keep if keep is not None else (locs if isinstance(locs, PtrTable) else list(locs))
        p, _k = engine._ptr_array(locs)
        kk = _u64_array(ks)
        h = c_vp()
        engine._check(engine._lib.flashe_span_bounds_create(engine._h, int(total), len(locs), p, ctypes.cast(kk, c_u64p), ctypes.byref(h)))
        self._h = h.value

    def recompute(self, locs, ks, keep=None):
        """The table for the next round's lists (same number of clients, same total), in place; mandatory after the lists changed."""
        p, _k = self.engine._ptr_array(locs)
        kk = _u64_array(ks)
        self.engine._check(self.engine._lib.flashe_span_bounds_recompute(self.engine._h, self._h, p, ctypes.cast(kk, c_u64p)))
        self._keep = keep if keep is not None else (locs if isinstance(locs, PtrTable) else list(locs))
        return self

    def __del__(self):
        try:
            if self._h:
                self.engine._lib.flashe_span_bounds_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Engine:
    """One device context of the cipher engine (wraps flashe_ctx)."""

    def __init__(self, key, int_bits, device=0, stream=None):
        self._lib = _lib.load()
        self._h = None
        key = bytes(key)
        if len(key) != 32:
            raise ValueError("key must be 32 bytes (AES-256)")
        h = c_vp()
        rc = self._lib.flashe_ctx_create(ctypes.byref(h), (ctypes.c_uint8 * 32).from_buffer_copy(key),
                                         int(int_bits), int(device), c_vp(stream) if stream else None)
        if rc:
            msg = self._lib.flashe_last_error(None)
            raise FlasheError(rc, msg.decode() if msg else "flashe_ctx_create failed")
        self._h = h.value
        self.int_bits = int(int_bits)
        self.limbs = limbs_of(int_bits)
        self.device = device

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc):
        return _lib.check(self._h, rc)

    def close(self):
        if self._h is not None:
            self._lib.flashe_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_key(self, key):
        key = bytes(key)
        assert len(key) == 32
        self._check(self._lib.flashe_ctx_set_key(self._h, (ctypes.c_uint8 * 32).from_buffer_copy(key)))

    def set_cu_limit(self, cus):
        """Persistent launches of this engine fill `cus` compute units (0 = all): leaves CUs free for kernels of another stream that
        cannot share a CU with a PRF workgroup (RCCL's)."""
        self._check(self._lib.flashe_ctx_set_cu_limit(self._h, int(cus)))

    @property
    def cu_count(self):
        return int(self._lib.flashe_ctx_cu_count(self._h))

    def set_prf_backend(self, backend):
        """0 = auto, 1 = LDS T-table kernel, 2 = bit-sliced VALU kernel (identical results)."""
        self._check(self._lib.flashe_ctx_set_prf_backend(self._h, int(backend)))

    def compact_supported(self):
        """True when the uint32 entry points (encrypt_batch_u32_dev, aggregate_decrypt_u32_dev, ...) run on this ctx: int_bits <= 32,
        the table PRF backend, the chained kernels.  The C side decides (flashe_ctx_compact_layout), so a caller never has to guess from
        environment variables what check_u32 will answer."""
        return int(self._lib.flashe_ctx_compact_layout(self._h)) == 1

    def selftest(self):
        self._check(self._lib.flashe_selftest(self._h))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def alloc_vec(self, n, limbs=None):
        return DeviceBuffer(self, max(int(n) * (limbs or self.limbs) * 8, 16))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 16)).upload(arr)

    def sync(self):
        self._check(self._lib.flashe_sync(self._h))

    def memset_dev(self, buf, byte, nbytes):
        """Fills the first nbytes of a device buffer with `byte` (asynchronous on the ctx stream)."""
        self._check(self._lib.flashe_memset_dev(self._h, self._ptr(buf), int(byte), int(nbytes)))

    # -- HIP graph capture / replay of a sequence of *_dev calls ---------------------------------
    def graph_begin(self):
        self._check(self._lib.flashe_graph_begin(self._h))

    def graph_end(self):
        g = c_vp()
        self._check(self._lib.flashe_graph_end(self._h, ctypes.byref(g)))
        return Graph(self, g)

    def event(self):
        ev = c_vp()
        self._check(self._lib.flashe_event_create(self._h, ctypes.byref(ev)))
        return ev.value

    def record(self, ev):
        self._check(self._lib.flashe_event_record(self._h, ev))

    def wait_event(self, ev):
        """Device-side wait of this engine's stream on an event recorded by another engine."""
        self._check(self._lib.flashe_stream_wait_event(self._h, ev))

    def elapsed_ms(self, start, stop):
        ms = ctypes.c_float(0)
        self._check(self._lib.flashe_event_elapsed_ms(self._h, start, stop, ctypes.byref(ms)))
        return float(ms.value)

    def event_destroy(self, ev):
        self._check(self._lib.flashe_event_destroy(self._h, ev))

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        return x.ptr if isinstance(x, (DeviceBuffer, DeviceVector)) else int(x)

    # -- device-pointer API (asynchronous on the ctx stream) -----------------------------------
    def mask_dev(self, it, idx_list, n, n_jobs, out):
        p, _keep = _u32_list(idx_list)
        self._check(self._lib.flashe_mask_dev(self._h, it, p, len(idx_list), n, n_jobs, self._ptr(out)))

    def encrypt_dev(self, it, idx, scheme, n, n_jobs, pt, pt_limbs, ct):
        self._check(self._lib.flashe_encrypt_dev(self._h, it, idx, scheme, n, n_jobs, self._ptr(pt), pt_limbs, self._ptr(ct)))

    def encrypt_batch_dev(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts):
        """len(idx_list) independent encrypts of equal length in as few launches as possible."""
        pi, _k = _u32_list(idx_list)
        pp, _a = self._ptr_array(pts)
        pc, _b = self._ptr_array(cts)
        self._check(self._lib.flashe_encrypt_batch_dev(self._h, it, scheme, n, n_jobs, len(idx_list), pi, pp, pt_limbs, pc))

    def encrypt_batch_sum_dev(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts, sum_out):
        """encrypt_batch_dev that also writes sum_out = sum of the ciphertexts mod 2^b (the local partial aggregate): one launch for
        a run of consecutive clients with int_bits > 64, the encrypts followed by the reduce otherwise."""
        pi, _k = _u32_list(idx_list)
        pp, _a = self._ptr_array(pts)
        pc, _b = self._ptr_array(cts)
        self._check(self._lib.flashe_encrypt_batch_sum_dev(self._h, it, scheme, n, n_jobs, len(idx_list), pi, pp, pt_limbs, pc,
                                                           self._ptr(sum_out)))

    def encrypt_batch_range_dev(self, it, idx_list, scheme, n, n_jobs, first, count, pts, pt_limbs, cts, sum_out=None):
        """encrypt_batch_dev on elements [first, first + count) of the n-element vectors -- the launch of a GPU that owns that slice of
        every client's vector (pointers address element `first`); sum_out (optional) receives the slice of the ciphertexts' sum."""
        pi, _k = _u32_list(idx_list)
        pp, _a = self._ptr_array(pts)
        pc, _b = self._ptr_array(cts)
        self._check(self._lib.flashe_encrypt_batch_range_dev(self._h, it, scheme, n, n_jobs, first, count, len(idx_list), pi, pp, pt_limbs, pc,
                                                             self._ptr(sum_out)))

    def prf_jobs_dev(self, it, n, n_jobs, jobs):
        """jobs: iterable of (add_idx, minus_idx or None, first, count, in_ptr or None, in_limbs, out_ptr); each writes
        out[k] = in[k] + term(it, add_idx, first + k) - term(it, minus_idx, first + k) for k < count (one launch for
        int_bits > 64).  Pointers address element `first`."""
        tab = jobs if isinstance(jobs, JobTable) else JobTable(self, jobs)
        self._check(self._lib.flashe_prf_jobs_dev(self._h, it, n, n_jobs, len(tab), tab.arr))

    def job_table(self, jobs):
        """The job list of prf_jobs_dev as a table built once (JobTable): for calls that repeat round after round."""
        return JobTable(self, jobs)

    # -- mask precompute resident in the ctx (FlasheCipher.prepare_encrypt / prepare_decrypt, jzf_flashe.py:599-666) ---------------
    PREPARED_ENCRYPT, PREPARED_DECRYPT = 1, 2

    def prepare_encrypt(self, it_next, idx, scheme, num_params, n_jobs):
        self._check(self._lib.flashe_prepare_encrypt(self._h, it_next, idx, scheme, num_params, n_jobs))

    def prepare_decrypt(self, it, num_clients, num_params, n_jobs):
        self._check(self._lib.flashe_prepare_decrypt(self._h, it, num_clients, num_params, n_jobs))

    def prepared_query(self, which):
        """(held?, n, add device pointer, minus device pointer) of the ctx's encrypt / decrypt mask cache."""
        n, a, m = c_u64(0), c_vp(), c_vp()
        rc = self._lib.flashe_prepared_query(self._h, which, ctypes.byref(n), ctypes.byref(a), ctypes.byref(m))
        if rc < 0:
            self._check(rc)
        return bool(rc), int(n.value), a.value, m.value

    def prepared_discard(self, which):
        self._check(self._lib.flashe_prepared_discard(self._h, which))

    def prepared_download(self, which, part):
        """The cached `add` / `minus` mask as a uint64 array [n, L] (None when nothing is held)."""
        held, n, a, m = self.prepared_query(which)
        ptr = a if part == "add" else m
        if not held or not ptr:
            return None
        out = host_empty((n, self.limbs), np.uint64)
        if n:
            self._check(self._lib.flashe_memcpy_d2h(self._h, out.ctypes.data, ptr, out.nbytes))
        return out

    def encrypt_prepared_dev(self, n, pt, pt_limbs, ct):
        """ct = pt + add - minus from the ctx's prepared encrypt masks (no AES); consumes them."""
        self._check(self._lib.flashe_encrypt_prepared_dev(self._h, n, self._ptr(pt), pt_limbs, self._ptr(ct)))

    def decrypt_prepared_dev(self, it, add_idx, minus_idx, n, n_jobs, inp, out):
        """out = inp + add - minus from the ctx's prepared decrypt masks, plus the listed extra prefixes computed online; consumes them."""
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        self._check(self._lib.flashe_decrypt_prepared_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs, self._ptr(inp), self._ptr(out)))

    def decrypt_dev(self, it, add_idx, minus_idx, n, n_jobs, inp, out):
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        self._check(self._lib.flashe_decrypt_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs,
                                                 self._ptr(inp), self._ptr(out)))

    def mask_range_dev(self, it, idx_list, n, n_jobs, first, count, out):
        p, _keep = _u32_list(idx_list)
        self._check(self._lib.flashe_mask_range_dev(self._h, it, p, len(idx_list), n, n_jobs, first, count, self._ptr(out)))

    def encrypt_range_dev(self, it, idx, scheme, n, n_jobs, first, count, pt, pt_limbs, ct):
        self._check(self._lib.flashe_encrypt_range_dev(self._h, it, idx, scheme, n, n_jobs, first, count,
                                                       self._ptr(pt), pt_limbs, self._ptr(ct)))

    def decrypt_range_dev(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out):
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        self._check(self._lib.flashe_decrypt_range_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs,
                                                       first, count, self._ptr(inp), self._ptr(out)))

    def combine_dev(self, n, inp, in_limbs, add, minus, out):
        self._check(self._lib.flashe_combine_dev(self._h, n, self._ptr(inp), in_limbs, self._ptr(add), self._ptr(minus), self._ptr(out)))

    def combine_batch_dev(self, n, inps, in_limbs, adds, minuses, outs):
        """out[v] = inps[v] + adds[v] - minuses[v] for every v in one launch; adds / minuses: lists (entries may be None) or None."""
        pi, _a = self._ptr_array(inps)
        po, _b = self._ptr_array(outs)
        pa, _c = self._ptr_array(adds) if adds is not None else (None, None)
        pm, _d = self._ptr_array(minuses) if minuses is not None else (None, None)
        self._check(self._lib.flashe_combine_batch_dev(self._h, n, len(inps), pi, in_limbs, pa, pm, po))

    def combine_batch_sum_dev(self, n, inps, in_limbs, adds, minuses, outs, sum_out):
        """combine_batch_dev and sum_out = sum_v outs[v] mod 2^b from the same pass (online encrypts with precomputed masks + the
        arbiter's reduce of what they wrote)."""
        pi, _a = self._ptr_array(inps)
        po, _b = self._ptr_array(outs)
        pa, _c = self._ptr_array(adds) if adds is not None else (None, None)
        pm, _d = self._ptr_array(minuses) if minuses is not None else (None, None)
        self._check(self._lib.flashe_combine_batch_sum_dev(self._h, n, len(inps), pi, in_limbs, pa, pm, po, self._ptr(sum_out)))

    def combine_batch_sum_decrypt_dev(self, n, inps, in_limbs, adds, minuses, outs, sum_out, dec_add, dec_minus, dec_out):
        """combine_batch_sum_dev and dec_out = (sum_out + dec_add - dec_minus) mod 2^b from the same pass: the online encrypts with
        precomputed masks, the arbiter's reduce of them and the decrypt of that reduce with the decrypting party's precomputed masks."""
        pi, _a = self._ptr_array(inps)
        po, _b = self._ptr_array(outs)
        pa, _c = self._ptr_array(adds) if adds is not None else (None, None)
        pm, _d = self._ptr_array(minuses) if minuses is not None else (None, None)
        self._check(self._lib.flashe_combine_batch_sum_decrypt_dev(self._h, n, len(inps), pi, in_limbs, pa, pm, po, self._ptr(sum_out),
                                                                   self._ptr(dec_add) if dec_add is not None else None,
                                                                   self._ptr(dec_minus) if dec_minus is not None else None, self._ptr(dec_out)))

    def _ptr_array(self, items):
        if isinstance(items, PtrTable):
            return items.ptr, items
        arr = (c_vp * max(len(items), 1))(*[self._ptr(x) for x in items])
        return ctypes.cast(arr, ctypes.POINTER(c_vp)), arr

    def ptr_table(self, items):
        """A list of device buffers as an argument table built once (PtrTable): accepted wherever a *_dev method takes such a list."""
        return PtrTable(self, items)

    @staticmethod
    def u64_table(values):
        return U64Table(values)

    def _zeros_array(self, zeros):
        """per client a sequence of L limbs (or an int) -> the flat limb array; a prebuilt ctypes array passes through"""
        if isinstance(zeros, ctypes.Array):
            return zeros
        flat = []
        for z in zeros:
            z = [int(z) & (2 ** 64 - 1), int(z) >> 64] if isinstance(z, int) else [int(v) for v in z] + [0]
            flat += z[:self.limbs]
        return (c_u64 * max(len(flat), 1))(*flat)

    def zeros_table(self, zeros):
        return self._zeros_array(zeros)

    def aggregate_elem_dev(self, cts, n, out):
        p, _keep = self._ptr_array(cts)
        self._check(self._lib.flashe_aggregate_elem_dev(self._h, len(cts), p, n, self._ptr(out)))

    def aggregate_decrypt_range_dev(self, it, add_idx, minus_idx, n, n_jobs, first, count, cts, agg_out, out):
        """agg = sum of cts mod 2^b (stored to agg_out unless None), out = decrypt of agg on elements [first, first + count);
        one pass when the ciphertexts are equally spaced in memory.  Pointers address element `first`."""
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        p, _keep = self._ptr_array(cts)
        self._check(self._lib.flashe_aggregate_decrypt_range_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs, first, count,
                                                                 len(cts), p, self._ptr(agg_out), self._ptr(out)))

    # -- compact layout for int_bits <= 32: the same values as uint32 arrays (half the bytes of the kernels bound by them) --------
    def encrypt_batch_u32_dev(self, it, idx_list, scheme, n, n_jobs, pts, cts):
        """encrypt_batch_dev on uint32 plaintext and ciphertext vectors (int_bits <= 32)."""
        pi, _k = _u32_list(idx_list)
        pp, _a = self._ptr_array(pts)
        pc, _b = self._ptr_array(cts)
        self._check(self._lib.flashe_encrypt_batch_u32_dev(self._h, it, scheme, n, n_jobs, len(idx_list), pi, pp, pc))

    def encrypt_batch_sum_u32_dev(self, it, idx_list, scheme, n, n_jobs, pts, cts, sum_out):
        """encrypt_batch_u32_dev and sum_out = sum of the ciphertexts (uint32 [n]) from the same launch where the shape allows it."""
        pi, _k = _u32_list(idx_list)
        pp, _a = self._ptr_array(pts)
        pc, _b = self._ptr_array(cts)
        self._check(self._lib.flashe_encrypt_batch_sum_u32_dev(self._h, it, scheme, n, n_jobs, len(idx_list), pi, pp, pc, self._ptr(sum_out)))

    def aggregate_decrypt_u32_dev(self, it, add_idx, minus_idx, n, n_jobs, first, count, cts, agg_out, out, out_elem_bytes=8):
        """aggregate_decrypt_range_dev on uint32 operands (one add, at most one minus prefix); agg_out / out are uint32 or uint64 arrays."""
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        p, _keep = self._ptr_array(cts)
        self._check(self._lib.flashe_aggregate_decrypt_u32_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs, first, count,
                                                               len(cts), p, self._ptr(agg_out), self._ptr(out), int(out_elem_bytes)))

    def aggregate_elem_u32_dev(self, cts, n, out):
        p, _keep = self._ptr_array(cts)
        self._check(self._lib.flashe_aggregate_elem_u32_dev(self._h, len(cts), p, n, self._ptr(out)))

    def widen_u32_dev(self, n, inp, out):
        self._check(self._lib.flashe_widen_u32_dev(self._h, n, self._ptr(inp), self._ptr(out)))

    def narrow_u32_dev(self, n, inp, out):
        self._check(self._lib.flashe_narrow_u32_dev(self._h, n, self._ptr(inp), self._ptr(out)))

    def aggregate_packed_dev(self, packed, n_limbs, total_bits, out):
        p, _keep = self._ptr_array(packed)
        self._check(self._lib.flashe_aggregate_packed_dev(self._h, len(packed), p, n_limbs, total_bits, self._ptr(out)))

    def packed_probe_dev(self, n_limbs, x, info):
        """info (device, 3 words) <- (x[0], body limbs [1, n_limbs-1) all ones, x[n_limbs-1]); asynchronous."""
        self._check(self._lib.flashe_packed_probe_dev(self._h, n_limbs, self._ptr(x), self._ptr(info)))

    def packed_add_carry_dev(self, n_limbs, total_bits, carry_in, x):
        """x <- (x + carry_in) mod 2^total_bits in place."""
        self._check(self._lib.flashe_packed_add_carry_dev(self._h, n_limbs, total_bits, carry_in, self._ptr(x)))

    def packed_resolve_carry_dev(self, n_limbs, total_bits, infos, n_below, x):
        """x <- (x + carry_in) mod 2^total_bits, carry_in derived on the device from the probe triples of the n_below slices below."""
        self._check(self._lib.flashe_packed_resolve_carry_dev(self._h, n_limbs, total_bits, self._ptr(infos), n_below, self._ptr(x)))

    def packed_resolve_carry_strided_dev(self, n_limbs, total_bits, infos, n_below, stride_words, x):
        self._check(self._lib.flashe_packed_resolve_carry_strided_dev(self._h, n_limbs, total_bits, self._ptr(infos), n_below, stride_words, self._ptr(x)))

    def pack_dev(self, n, inp, out):
        self._check(self._lib.flashe_pack_dev(self._h, n, self._ptr(inp), self._ptr(out)))

    def unpack_dev(self, n, inp, out):
        self._check(self._lib.flashe_unpack_dev(self._h, n, self._ptr(inp), self._ptr(out)))

    def expand_to_dense_dev(self, total, k, loc, vals, zero_limbs, out):
        z = (c_u64 * 2)(*([int(v) for v in zero_limbs] + [0])[:2])
        self._check(self._lib.flashe_expand_to_dense_dev(self._h, total, k, self._ptr(loc), self._ptr(vals),
                                                         ctypes.cast(z, c_u64p), self._ptr(out)))

    def span_bounds(self, total, locs, ks, keep=None):
        """The span bounds of a round's strictly increasing location lists, computed once: a SpanBounds handle that
        sparse_aggregate_dev / sparse_decrypt_dev take (bounds=...) instead of recomputing the table for the same lists.  After an
        in-place change of a list: handle.recompute(...) first.  keep: objects the handle should keep alive (locs given as addresses)."""
        return SpanBounds(self, total, locs, ks, keep=keep)

    def sparse_aggregate_dev(self, total, locs, ks, vals, zeros, out, sorted_lists=False, bounds=None):
        """out = sum over clients of expand_to_dense(total, locs[c], vals[c], zeros[c]) mod 2^b, without the dense
        intermediates.  zeros: per client a sequence of L limbs (or an int).  sorted_lists: every location list is
        strictly increasing (one-pass LDS-staged form).  bounds: a SpanBounds of exactly these lists (implies sorted_lists)."""
        pl, _kl = self._ptr_array(locs)
        pv, _kv = self._ptr_array(vals)
        k = _u64_array(ks)
        zz = self._zeros_array(zeros)
        if bounds is not None:
            self._check(self._lib.flashe_sparse_aggregate_bounds_dev(self._h, total, len(locs), pl, ctypes.cast(k, c_u64p), pv,
                                                                     ctypes.cast(zz, c_u64p), bounds._h, self._ptr(out)))
            return
        self._check(self._lib.flashe_sparse_aggregate_dev(self._h, total, len(locs), pl, ctypes.cast(k, c_u64p), pv,
                                                          ctypes.cast(zz, c_u64p), 1 if sorted_lists else 0, self._ptr(out)))

    def sparse_encrypt_aggregate_dev(self, it, idx, locs, ks, pts, pt_limbs, zeros, total, n_jobs, cts, agg, bounds=None, position_range=None):
        """The clients this device plays encrypt their compact uploads (single mask) and the sum of the expanded uploads is written in
        the same pass: cts[c] = encrypt(it, idx[c], SINGLE) of pts[c], agg = sparse_aggregate_dev(locs, cts, zeros).  Strictly increasing
        location lists.  int_bits > 64: one persistent launch per 64 clients with the PRF inside the span reduce.
        position_range = (first, count): only the positions [first, first + count) -- one GPU's share of a round sharded by position ranges;
        `agg` then addresses position `first`, the ciphertexts of entries outside the range are not written."""
        pl, _kl = self._ptr_array(locs)
        pp, _kp = self._ptr_array(pts)
        pc, _kc = self._ptr_array(cts)
        C = len(locs)
        k = _u64_array(ks)
        ii = idx if isinstance(idx, ctypes.Array) else (ctypes.c_uint32 * max(C, 1))(*[int(v) for v in idx])
        zz = self._zeros_array(zeros)
        if position_range is not None:
            first, count = position_range
            if bounds is None:
                raise ValueError("a position range needs the span bounds of the lists")
            self._check(self._lib.flashe_sparse_encrypt_aggregate_range_dev(self._h, it, n_jobs, total, C, ii, pl, ctypes.cast(k, c_u64p), pp, pt_limbs,
                                                                            ctypes.cast(zz, c_u64p), bounds._h, int(first), int(count), pc, self._ptr(agg)))
            return
        self._check(self._lib.flashe_sparse_encrypt_aggregate_dev(self._h, it, n_jobs, total, C, ii, pl, ctypes.cast(k, c_u64p), pp, pt_limbs,
                                                                  ctypes.cast(zz, c_u64p), bounds._h if bounds is not None else None, pc,
                                                                  self._ptr(agg)))

    def sparse_span(self):
        """Positions per span of the sparse passes with the PRF inside: position ranges start (and, unless they end the vector, end) at
        multiples of it."""
        return int(self._lib.flashe_sparse_span())

    def sparse_minus_mask_dev(self, it, locs, ks, total, n_jobs, out, sorted_lists=False):
        p, _keep = self._ptr_array(locs)
        k = _u64_array(ks)
        fn = self._lib.flashe_sparse_minus_mask_sorted_dev if sorted_lists else self._lib.flashe_sparse_minus_mask_dev
        self._check(fn(self._h, it, len(locs), p, ctypes.cast(k, c_u64p), total, n_jobs, self._ptr(out)))

    def sparse_decrypt_dev(self, it, locs, ks, total, n_jobs, agg, out, sorted_lists=False, bounds=None, position_range=None):
        """out = (agg - dense minus-mask of the location lists) mod 2^b in the pass that builds the mask.  position_range = (first, count):
        agg / out address position `first` and hold `count` elements (needs bounds)."""
        p, _keep = self._ptr_array(locs)
        k = _u64_array(ks)
        if position_range is not None:
            first, count = position_range
            if bounds is None:
                raise ValueError("a position range needs the span bounds of the lists")
            self._check(self._lib.flashe_sparse_decrypt_range_dev(self._h, it, len(locs), p, ctypes.cast(k, c_u64p), total, n_jobs, bounds._h,
                                                                  int(first), int(count), self._ptr(agg), self._ptr(out)))
            return
        if bounds is not None:
            self._check(self._lib.flashe_sparse_decrypt_bounds_dev(self._h, it, len(locs), p, ctypes.cast(k, c_u64p), total, n_jobs, bounds._h,
                                                                   self._ptr(agg), self._ptr(out)))
            return
        self._check(self._lib.flashe_sparse_decrypt_dev(self._h, it, len(locs), p, ctypes.cast(k, c_u64p), total, n_jobs,
                                                        1 if sorted_lists else 0, self._ptr(agg), self._ptr(out)))

    def sparse_double_masks_dev(self, it, locs, ks, total, add_out, minus_out):
        """Both dense masks of the sparse + double decrypt (set_idx_list's run analysis + _static_prepare_decrypt_spar) straight from
        the clients' strictly increasing location lists."""
        p, _keep = self._ptr_array(locs)
        k = (c_u64 * max(len(ks), 1))(*[int(v) for v in ks])
        self._check(self._lib.flashe_sparse_double_masks_dev(self._h, it, len(locs), p, ctypes.cast(k, c_u64p), total,
                                                             self._ptr(add_out), self._ptr(minus_out)))

    def dynamic_masking_cost_dev(self, locs, ks):
        """(single_cost, double_cost) of Arbiter.dynamic_masking (jzf_flashe_block.py:92-112) from device-resident, strictly increasing
        location lists: the positions consecutive clients share are counted on the device, no one-hot vectors."""
        p, _keep = self._ptr_array(locs)
        kk = (c_u64 * max(len(ks), 1))(*[int(v) for v in ks])
        single, double = c_u64(0), c_u64(0)
        self._check(self._lib.flashe_dynamic_masking_cost_dev(self._h, len(locs), p, kk, ctypes.byref(single), ctypes.byref(double)))
        return int(single.value), int(double.value)

    def sparse_dense_mask_dev(self, it, sels, total, out):
        p, _keep = self._ptr_array(sels)
        self._check(self._lib.flashe_sparse_dense_mask_dev(self._h, it, len(sels), p, total, self._ptr(out)))

    # -- host-array API (synchronous; numpy uint64 limb arrays in and out) -----------------------
    def _vec(self, arr, allow_pt=False):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        if arr.shape[1] != self.limbs and not (allow_pt and arr.shape[1] == 1):
            raise ValueError(f"expected [n, {self.limbs}] uint64 limbs, got {arr.shape}")
        return arr

    def mask(self, it, idx_list, n, n_jobs):
        out = host_empty((n, self.limbs))
        p, _keep = _u32_list(idx_list)
        self._check(self._lib.flashe_mask(self._h, it, p, len(idx_list), n, n_jobs, out.ctypes.data))
        return out

    def encrypt(self, it, idx, scheme, n_jobs, pt):
        pt = self._vec(pt, allow_pt=True)
        n = pt.shape[0]
        ct = host_empty((n, self.limbs))
        self._check(self._lib.flashe_encrypt(self._h, it, idx, scheme, n, n_jobs, pt.ctypes.data, pt.shape[1], ct.ctypes.data))
        return ct

    def decrypt(self, it, add_idx, minus_idx, n_jobs, ct):
        ct = self._vec(ct)
        n = ct.shape[0]
        out = host_empty(ct.shape)
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        self._check(self._lib.flashe_decrypt(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs,
                                             ct.ctypes.data, out.ctypes.data))
        return out

    def combine(self, inp, add=None, minus=None):
        inp = self._vec(inp, allow_pt=True)
        n = inp.shape[0]
        add = self._vec(add) if add is not None else None
        minus = self._vec(minus) if minus is not None else None
        out = host_empty((n, self.limbs))
        self._check(self._lib.flashe_combine(self._h, n, inp.ctypes.data, inp.shape[1],
                                             add.ctypes.data if add is not None else None,
                                             minus.ctypes.data if minus is not None else None, out.ctypes.data))
        return out

    def aggregate_elem(self, cts):
        cts = [self._vec(c) for c in cts]
        n = cts[0].shape[0]
        if any(c.shape[0] != n for c in cts):
            raise ValueError("operands differ in length")
        out = host_empty((n, self.limbs))
        tab = (c_vp * len(cts))(*[c.ctypes.data for c in cts])
        self._check(self._lib.flashe_aggregate_elem(self._h, len(cts), ctypes.cast(tab, ctypes.POINTER(c_vp)), n, out.ctypes.data))
        return out

    def aggregate_packed(self, packed, total_bits):
        n_limbs = (total_bits + 63) // 64
        packed = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1) for p in packed]
        if any(p.shape[0] != n_limbs for p in packed):
            raise ValueError("operands must have ceil(total_bits / 64) limbs")
        out = np.zeros(n_limbs, dtype=np.uint64)
        tab = (c_vp * len(packed))(*[p.ctypes.data for p in packed])
        self._check(self._lib.flashe_aggregate_packed(self._h, len(packed), ctypes.cast(tab, ctypes.POINTER(c_vp)),
                                                      n_limbs, total_bits, out.ctypes.data))
        return out

    def pack(self, x):
        x = self._vec(x)
        n = x.shape[0]
        out = np.zeros((n * self.int_bits + 63) // 64, dtype=np.uint64)
        self._check(self._lib.flashe_pack(self._h, n, x.ctypes.data, out.ctypes.data))
        return out

    def unpack(self, p, n):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1)
        if p.shape[0] != (n * self.int_bits + 63) // 64:
            raise ValueError("packed operand has the wrong number of limbs")
        out = host_empty((n, self.limbs))
        self._check(self._lib.flashe_unpack(self._h, n, p.ctypes.data, out.ctypes.data))
        return out

    def expand_to_dense(self, total, loc, vals, zero_limbs):
        loc = np.ascontiguousarray(loc, dtype=np.uint32)
        vals = self._vec(vals) if len(loc) else np.zeros((0, self.limbs), dtype=np.uint64)
        z = (c_u64 * 2)(*([int(v) for v in np.asarray(zero_limbs).reshape(-1)] + [0])[:2])
        out = host_empty((total, self.limbs))
        self._check(self._lib.flashe_expand_to_dense(self._h, total, len(loc), loc.ctypes.data, vals.ctypes.data,
                                                     ctypes.cast(z, c_u64p), out.ctypes.data))
        return out

    def sparse_minus_mask(self, it, locs, total, n_jobs):
        locs = [np.ascontiguousarray(l, dtype=np.uint32) for l in locs]
        tab = (c_vp * max(len(locs), 1))(*[l.ctypes.data for l in locs])
        k = (c_u64 * max(len(locs), 1))(*[len(l) for l in locs])
        out = host_empty((total, self.limbs))
        self._check(self._lib.flashe_sparse_minus_mask(self._h, it, len(locs), ctypes.cast(tab, ctypes.POINTER(c_vp)),
                                                       ctypes.cast(k, c_u64p), total, n_jobs, out.ctypes.data))
        return out

    def sparse_dense_mask(self, it, sels, total):
        sels = [np.ascontiguousarray(s, dtype=np.uint8) for s in sels]
        if any(s.shape[0] != total for s in sels):
            raise ValueError("selector length must equal total")
        tab = (c_vp * max(len(sels), 1))(*[s.ctypes.data for s in sels])
        out = host_empty((total, self.limbs))
        self._check(self._lib.flashe_sparse_dense_mask(self._h, it, len(sels), ctypes.cast(tab, ctypes.POINTER(c_vp)),
                                                       total, out.ctypes.data))
        return out

    # -- quantise / batch codec (jzf_quantize.py *_padding_asymmetric) ----------------------------
    def quantize_dev(self, n, x, x_is_f64, alpha, element_bits, u, q):
        self._check(self._lib.flashe_quantize_dev(self._h, n, self._ptr(x), 1 if x_is_f64 else 0, float(alpha), element_bits,
                                                  self._ptr(u), self._ptr(q)))

    def unquantize_dev(self, n, v, v_limbs, alpha, element_bits, num_clients, out):
        self._check(self._lib.flashe_unquantize_dev(self._h, n, self._ptr(v), v_limbs, float(alpha), element_bits, num_clients,
                                                    self._ptr(out)))

    def batch_dev(self, n, vals, field_bits, out):
        self._check(self._lib.flashe_batch_dev(self._h, n, self._ptr(vals), field_bits, self._ptr(out)))

    def unbatch_dev(self, n_batches, inp, field_bits, out):
        self._check(self._lib.flashe_unbatch_dev(self._h, n_batches, self._ptr(inp), field_bits, self._ptr(out)))

    def numpy_random_dev(self, n, out=None):
        """np.random.random(n) generated on the device, bit for bit, from NumPy's GLOBAL legacy generator: its MT19937 state is read,
        advanced on the device and put back, so host draws before and after continue one stream.  Returns the DeviceBuffer of n
        float64 (`out` or a new one).  Raises if the global generator is not MT19937."""
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise FlasheError(-22, f"np.random's bit generator is {st[0]}, not MT19937")
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = c_u32(int(st[2]))
        out = out if out is not None else self.alloc(max(8 * int(n), 16))
        self._check(self._lib.flashe_mt19937_random_dev(self._h, key.ctypes.data_as(c_u32p), ctypes.byref(pos), int(n), self._ptr(out)))
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
        return out

    def quantize_encrypt_dev(self, it, idx, scheme, n, n_jobs, x, x_is_f64, alpha, element_bits, u, ct):
        """ct = encrypt(quantize(x)) in one launch (un-batched values); x float32 / float64, u float64 uniforms, all device-resident."""
        self._check(self._lib.flashe_quantize_encrypt_dev(self._h, it, idx, scheme, n, n_jobs, self._ptr(x), 1 if x_is_f64 else 0,
                                                          float(alpha), element_bits, self._ptr(u), self._ptr(ct)))

    def decrypt_unquantize_dev(self, it, add_idx, minus_idx, n, n_jobs, inp, alpha, element_bits, num_clients, out):
        """out (float64) = unquantize(decrypt(inp)) in one launch."""
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        self._check(self._lib.flashe_decrypt_unquantize_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs, self._ptr(inp),
                                                            float(alpha), element_bits, num_clients, self._ptr(out)))

    @staticmethod
    def _codec_layers(layers):
        """layers: iterable of (start, x device pointer or None, alpha, x_is_f64) -> ctypes array of flashe_codec_layer."""
        layers = list(layers)
        arr = (_lib.CodecLayer * max(len(layers), 1))()
        for i, (start, x, alpha, is_f64) in enumerate(layers):
            arr[i].start, arr[i].x_dev, arr[i].alpha, arr[i].x_is_f64, arr[i].reserved = int(start), x, float(alpha), 1 if is_f64 else 0, 0
        return arr, len(layers)

    def quantize_encrypt_model_dev(self, it, idx, scheme, n, n_jobs, first, count, layers, element_bits, u, ct):
        """The fused quantise -> encrypt over elements [first, first + count) of a FLATTENED model of n values (PRF counters and, for
        int_bits <= 64, the chunking run across the layers, as in a reference job: jzf_aggregator.py:721-741).  layers: (start, device
        pointer of the layer's own first value, alpha, is_f64) in ascending start order; u / ct address element `first`."""
        arr, nl = self._codec_layers(layers)
        self._check(self._lib.flashe_quantize_encrypt_model_dev(self._h, it, idx, scheme, n, n_jobs, first, count, arr, nl, element_bits,
                                                                self._ptr(u), self._ptr(ct)))

    def decrypt_unquantize_model_dev(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, layers, element_bits, num_clients, out):
        """The fused decrypt -> unquantise of the same flattened vector: layers = (start, None, alpha, False) per layer."""
        pa, _a = _u32_list(add_idx)
        pm, _m = _u32_list(minus_idx)
        arr, nl = self._codec_layers(layers)
        self._check(self._lib.flashe_decrypt_unquantize_model_dev(self._h, it, pa, len(add_idx), pm, len(minus_idx), n, n_jobs, first, count,
                                                                  self._ptr(inp), arr, nl, element_bits, num_clients, self._ptr(out)))

    def unquantize_model_dev(self, n, first, count, inp, layers, element_bits, num_clients, out):
        """The codec back end alone over elements [first, first + count) of a flattened, already decrypted vector (the sparse job's way back)."""
        arr, nl = self._codec_layers(layers)
        self._check(self._lib.flashe_unquantize_model_dev(self._h, n, first, count, self._ptr(inp), arr, nl, element_bits, num_clients, self._ptr(out)))

    @staticmethod
    def _batch_layers(layers):
        """layers: iterable of (size, x device pointer or None, alpha, x_is_f64) -> ctypes array of flashe_batch_layer."""
        layers = list(layers)
        arr = (_lib.BatchLayer * max(len(layers), 1))()
        for i, (size, x, alpha, is_f64) in enumerate(layers):
            arr[i].size, arr[i].x_dev, arr[i].alpha, arr[i].x_is_f64, arr[i].reserved = int(size), x, float(alpha), 1 if is_f64 else 0, 0
        return arr, len(layers)

    def quantize_batch_model_dev(self, layers, element_bits, field_bits, u, n_elems, out):
        """The batched plaintext of a whole model in one launch: every layer quantised with its own alpha and batched on its own
        (int_bits // field_bits values per element, zero padded), the batched layers back to back in `out` (n_elems x L limbs)."""
        arr, nl = self._batch_layers(layers)
        self._check(self._lib.flashe_quantize_batch_model_dev(self._h, arr, nl, element_bits, field_bits, self._ptr(u), n_elems, self._ptr(out)))

    def unbatch_unquantize_model_dev(self, layers, element_bits, field_bits, num_clients, inp, n_elems, out):
        """The way back: the decrypted flattened batched vector -> the model's float64 values in walking order, one launch."""
        arr, nl = self._batch_layers(layers)
        self._check(self._lib.flashe_unbatch_unquantize_model_dev(self._h, arr, nl, element_bits, field_bits, num_clients, self._ptr(inp), n_elems,
                                                                  self._ptr(out)))

    def shift_dev(self, n, x, x_is_f64, shift, wide=False):
        """x <- x + shift in place (normalize: shift = -mean)."""
        self._check(self._lib.flashe_shift_dev(self._h, n, self._ptr(x), 1 if x_is_f64 else 0, float(shift), 1 if wide else 0))

    def mean_std_dev(self, n, x, x_is_f64):
        """(mean, std) of a device vector, float64 two-pass reduction (synchronous)."""
        m, s = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.flashe_mean_std_dev(self._h, n, self._ptr(x), 1 if x_is_f64 else 0, ctypes.byref(m), ctypes.byref(s)))
        return float(m.value), float(s.value)

    def quantize(self, x, alpha, element_bits, u):
        x = np.ascontiguousarray(x)
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1)
        x = x.reshape(-1)
        if u.shape[0] != x.shape[0]:
            raise ValueError("need one uniform draw per element")
        q = np.zeros(x.shape[0], dtype=np.uint64)
        self._check(self._lib.flashe_quantize(self._h, x.shape[0], x.ctypes.data, 1 if x.dtype == np.float64 else 0,
                                              float(alpha), element_bits, u.ctypes.data, q.ctypes.data))
        return q

    def unquantize(self, v, alpha, element_bits, num_clients):
        v = np.ascontiguousarray(v, dtype=np.uint64)
        if v.ndim == 1:
            v = v.reshape(-1, 1)
        out = np.zeros(v.shape[0], dtype=np.float64)
        self._check(self._lib.flashe_unquantize(self._h, v.shape[0], v.ctypes.data, v.shape[1], float(alpha), element_bits,
                                                num_clients, out.ctypes.data))
        return out

    def batch(self, vals, field_bits):
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1)
        bs = self.int_bits // field_bits
        nb = (vals.shape[0] + bs - 1) // bs
        out = np.zeros((nb, self.limbs), dtype=np.uint64)
        self._check(self._lib.flashe_batch(self._h, vals.shape[0], vals.ctypes.data, field_bits, out.ctypes.data))
        return out

    def unbatch(self, batched, field_bits):
        batched = self._vec(batched)
        bs = self.int_bits // field_bits
        out = np.zeros(batched.shape[0] * bs, dtype=np.uint64)
        self._check(self._lib.flashe_unbatch(self._h, batched.shape[0], batched.ctypes.data, field_bits, out.ctypes.data))
        return out

    # -- top-k sparsifier (Client.sparsify, jzf_aggregator.py:578-623) ---------------------------------
    def sparsify_dev(self, n, k, x, x_is_f64, residual, loc, vals):
        self._check(self._lib.flashe_sparsify_dev(self._h, n, k, self._ptr(x), 1 if x_is_f64 else 0, self._ptr(residual),
                                                  self._ptr(loc), self._ptr(vals)))

    def sparsify_batch_dev(self, ns, ks, x, x_is_f64, residual, loc, vals):
        """Top-k of every layer of a model in one set of launches: layers back to back in the flat device vectors x / residual, outputs
        back to back in loc / vals (locations relative to their layer)."""
        L = len(ns)
        an = (c_u64 * max(L, 1))(*[int(v) for v in ns])
        ak = (c_u64 * max(L, 1))(*[int(v) for v in ks])
        self._check(self._lib.flashe_sparsify_batch_dev(self._h, L, an, ak, self._ptr(x), 1 if x_is_f64 else 0, self._ptr(residual),
                                                        self._ptr(loc), self._ptr(vals)))

    def sparsify_batch(self, layers, ks, residuals=None):
        """[(loc uint32[k_l] ascending, vals[k_l], new residual or None) per layer] -- Client.sparsify's layer loop
        (jzf_aggregator.py:585-613) as ONE upload, one set of launches and one download.  All layers share one float type."""
        flats = [np.ascontiguousarray(l).reshape(-1) for l in layers]
        dt = np.result_type(*[f.dtype for f in flats]) if flats else np.float64
        if dt not in (np.float32, np.float64):
            dt = np.float64
        ns = [int(f.size) for f in flats]
        ks = [int(v) for v in ks]
        x = np.concatenate([f.astype(dt, copy=False) for f in flats]) if flats else np.zeros(0, dtype=dt)
        res = None
        if residuals is not None:
            res = np.concatenate([np.ascontiguousarray(r, dtype=dt).reshape(-1) for r in residuals]) if flats else np.zeros(0, dtype=dt)
        loc = np.zeros(sum(ks), dtype=np.uint32)
        vals = np.zeros(sum(ks), dtype=dt)
        L = len(ns)
        an = (c_u64 * max(L, 1))(*ns)
        ak = (c_u64 * max(L, 1))(*ks)
        self._check(self._lib.flashe_sparsify_batch(self._h, L, an, ak, x.ctypes.data, 1 if dt == np.float64 else 0,
                                                    res.ctypes.data if res is not None else None, loc.ctypes.data, vals.ctypes.data))
        out, o, q = [], 0, 0
        for n_l, k_l in zip(ns, ks):
            out.append((loc[q:q + k_l], vals[q:q + k_l], None if res is None else res[o:o + n_l]))
            o += n_l
            q += k_l
        return out

    def sparsify_model(self, layers, ks, residual_dev=None, dtype=None):
        """sparsify_batch for a caller that keeps the residuals ON THE DEVICE between rounds: every layer is copied straight into its
        slot of one flat device buffer (no host-side concatenation), residual_dev (a DeviceBuffer of the flat residuals, updated in
        place; None = no residual) never crosses PCIe, only the k_l selected entries come back.  -> [(loc, vals) per layer]."""
        flats = [np.ascontiguousarray(l).reshape(-1) for l in layers]
        dt = np.dtype(dtype) if dtype is not None else (np.result_type(*[f.dtype for f in flats]) if flats else np.dtype(np.float64))
        if dt not in (np.float32, np.float64):
            dt = np.dtype(np.float64)
        ns = [int(f.size) for f in flats]
        ks = [int(v) for v in ks]
        total, total_k = sum(ns), sum(ks)
        dx = self.alloc(max(total * dt.itemsize, 16))
        off = 0
        for f in flats:
            f = f.astype(dt, copy=False)
            if f.size:
                self._check(self._lib.flashe_memcpy_h2d(self._h, dx.ptr + off, f.ctypes.data, f.nbytes))
            off += f.nbytes
        dl, dv = self.alloc(max(4 * total_k, 16)), self.alloc(max(dt.itemsize * total_k, 16))
        self.sparsify_batch_dev(ns, ks, dx, dt == np.float64, residual_dev, dl, dv)
        loc, vals = dl.download(np.uint32, total_k), dv.download(dt, total_k)
        out, q = [], 0
        for k_l in ks:
            out.append((loc[q:q + k_l], vals[q:q + k_l]))
            q += k_l
        return out

    def sparsify(self, layer, k, residual=None):
        """-> (loc uint32[k] ascending, vals[k] = layer + residual at loc, new residual or None)."""
        layer = np.ascontiguousarray(layer).reshape(-1)
        if layer.dtype not in (np.float32, np.float64):
            layer = layer.astype(np.float64)
        res = None if residual is None else np.ascontiguousarray(residual, dtype=layer.dtype).reshape(-1).copy()
        loc = np.zeros(k, dtype=np.uint32)
        vals = np.zeros(k, dtype=layer.dtype)
        self._check(self._lib.flashe_sparsify(self._h, layer.shape[0], k, layer.ctypes.data, 1 if layer.dtype == np.float64 else 0,
                                              res.ctypes.data if res is not None else None, loc.ctypes.data, vals.ctypes.data))
        return loc, vals, res
