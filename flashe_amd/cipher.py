"""`FlasheCipher` -- host-side mirror of the reference class of the same name
(federatedml/secureprotol/jzf_flashe.py:228-666) on top of the MI355X engine.

Same constructor, methods, attributes, state machine and error behaviour as the
reference, so the adapter that drives it (jzf_flashe_block._Client, jzf_flashe_block.py:
142-174) and the notebook harness (encrypt_test/final_big_table.ipynb cell 14) can switch
by changing one import.  What differs is where the arithmetic runs: every mask stream,
encrypt, decrypt and aggregate goes through libflashe_hip.so (HIP kernels); nothing is
computed in Python and there is no CPU fallback.

Data convention: like the reference, `encrypt` / `decrypt` take a 1-D
``np.ndarray(dtype=object)`` of Python ints and return a NEW array of the same kind.
As a fast path they also accept ``uint64`` arrays -- shape ``[n]`` (values < 2**64) or
``[n, L]`` little-endian limbs, L = ceil(int_bits / 64) -- and then return ``[n, L]``
(``[n]`` when L == 1 and the input was 1-D) without ever building Python ints.

Device-resident mode (new): ``encrypt`` / ``decrypt`` / ``aggregate`` take ``device=True`` and
then return a `DeviceVector` (the result stays in HBM; ``.to_host()`` downloads it), and they
accept a `DeviceVector` wherever they accept an array -- a `DeviceVector` input makes the
output one too unless ``device=False``.  A round then uploads each plaintext once and
downloads one result, instead of bouncing every ciphertext device -> host -> device between
the three calls.  The reference's type check is kept: anything that is neither an ndarray nor
a `DeviceVector` gives ``None``.

`N_JOBS` mirrors the reference's module global (jzf_flashe.py:7): for int_bits <= 64 the
PRF counters depend on chunks_idx(range(n), N_JOBS), so every party must use the same
value (the reference uses cpu_count(); set `flashe_amd.cipher.N_JOBS` to match a peer).
"""
import os
from multiprocessing import cpu_count

import numpy as np

from . import engine as _engine
from .engine import SCHEME_DOUBLE, SCHEME_SINGLE, DeviceVector, Engine

import logging
import time

N_JOBS = cpu_count()
# Phase markers: the reference brackets its phases with LOGGER.info("start encoding") / ("end encoding")
# lines that its log post-processing turns into per-phase times.  The mirror logs one line per phase on this logger
# (DEBUG level: silent unless enabled) -- name, vector length and wall time.
LOGGER = logging.getLogger("flashe_amd.cipher")
BITS_PER_BYTES = 8
_M64 = (1 << 64) - 1

__all__ = ["FlasheCipher", "aggregate", "DeviceVector", "N_JOBS"]


# ------------------------------------------------------------------------------ conversions
def _load_pyconv():
    """flashe_amd/_pyconv.so (csrc/pyconv.c): object array <-> limbs through the CPython API, 4-6x NumPy's astype().  Host-side
    format conversion only; when it is not built the NumPy path below gives the same arrays."""
    import ctypes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pyconv.so")
    if not os.path.exists(path) or os.environ.get("FLASHE_NO_PYCONV"):
        return None
    try:
        lib = ctypes.PyDLL(path)
        for fn in (lib.flashe_pyconv_ints_to_limbs, lib.flashe_pyconv_limbs_to_ints):
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_void_p]
        return lib
    except (OSError, AttributeError):
        return None


_PYCONV = _load_pyconv()


def _to_limbs(value, limbs):
    """ndarray (object ints | uint64) -> (uint64 [n, k] array, kind) with k in {1, limbs}."""
    if value.dtype == object:
        if value.ndim != 1:
            value = value.reshape(-1)
        n = value.shape[0]
        out = np.empty((n, limbs), dtype=np.uint64)
        if n and _PYCONV is not None:
            src = np.ascontiguousarray(value)
            # raises the Python error the conversion hit (TypeError for a non-integer element, as int() would)
            _PYCONV.flashe_pyconv_ints_to_limbs(src.ctypes.data, n, limbs, out.ctypes.data)
            return out, "object"
        if n:
            try:
                # the usual case -- non-negative values below 2**64 -- converts in C loops.  A negative value must NOT take
                # this path: the reference reduces it mod 2**int_bits (sign-extended high limb), and astype(uint64) would wrap
                # a negative NumPy scalar held in the object array without raising.
                try:
                    low = value.astype(np.int64)                 # raises beyond int64
                    if (low < 0).any():
                        raise OverflowError("negative value")
                    out[:, 0] = low.view(np.uint64)
                except OverflowError:
                    low = value.astype(np.uint64)                # values in [2**63, 2**64); raises beyond
                    if (value.astype(np.float64) < 0).any():
                        raise OverflowError("negative value")
                    out[:, 0] = low
                if limbs == 2:
                    out[:, 1] = 0
            except (OverflowError, TypeError):
                # the general path works on Python ints (NumPy scalars held in the object array would overflow in `&`)
                value = np.fromiter((int(v) for v in value), dtype=object, count=n)
                out[:, 0] = (value & _M64).astype(np.uint64)
                if limbs == 2:
                    out[:, 1] = ((value >> 64) & _M64).astype(np.uint64)
        return out, "object"
    if value.dtype == np.uint64:
        if value.ndim == 1:
            return np.ascontiguousarray(value).reshape(-1, 1), "u64_1d"
        if value.ndim == 2 and value.shape[1] in (1, limbs):
            return np.ascontiguousarray(value), "u64_2d"
        raise ValueError(f"uint64 input must be [n] or [n, {limbs}], got {value.shape}")
    if np.issubdtype(value.dtype, np.integer):
        return _to_limbs(value.astype(object), limbs)
    raise TypeError(f"unsupported dtype {value.dtype}: pass object ints or uint64 limbs")


def _from_limbs(arr, kind):
    if kind == "object" and _PYCONV is not None:
        src = np.ascontiguousarray(arr, dtype=np.uint64)
        out = np.empty(src.shape[0], dtype=object)
        if src.shape[0]:
            _PYCONV.flashe_pyconv_limbs_to_ints(src.ctypes.data, src.shape[0], src.shape[1], out.ctypes.data)
        return out
    if kind == "object":
        out = arr[:, 0].astype(object)
        if arr.shape[1] == 2:
            out = out | (arr[:, 1].astype(object) << 64)
        return out
    if kind == "u64_1d" and arr.shape[1] == 1:
        return arr[:, 0]
    return arr


class _DevVec:
    """A mask vector resident in HBM (the cached `next_iter_*_prepared` entries)."""

    def __init__(self, eng, n):
        self.n = n
        self.buf = eng.alloc_vec(n)

    def __len__(self):
        return self.n

    def to_host(self, eng):
        return self.buf.download(np.uint64, self.n * eng.limbs).reshape(self.n, eng.limbs)


class _SparseMinus:
    """The single-mask sparse branch of set_idx_list (jzf_flashe.py:316-343) as a promise: the clients' location lists are on the
    device, the dense minus-mask itself is not built unless somebody asks for it (`buf` / `to_host`).  The decrypt that follows
    subtracts the mask from the aggregate in the pass that computes it (flashe_sparse_decrypt_dev: one read and one write of the dense
    vector instead of building 16 B x total, reading it back and combining)."""

    def __init__(self, eng, it, dloc, ks, total, sorted_lists):
        self.eng, self.it, self.dloc, self.ks, self.n, self.sorted_lists = eng, it, dloc, ks, total, sorted_lists
        self._vec = None

    def __len__(self):
        return self.n

    @property
    def buf(self):
        if self._vec is None:
            self._vec = _DevVec(self.eng, self.n)
            self.eng.sparse_minus_mask_dev(self.it, self.dloc, self.ks, self.n, N_JOBS, self._vec.buf, sorted_lists=self.sorted_lists)
            self.eng.sync()
        return self._vec.buf

    @property
    def materialized(self):
        return self._vec is not None

    def to_host(self, eng):
        _ = self.buf
        return self._vec.to_host(eng)


class _CtxMask:
    """A precomputed mask held INSIDE the engine's ctx (flashe_prepare_encrypt / flashe_prepare_decrypt): the entry the reference keeps
    in next_iter_encrypt_prepared / next_iter_decrypt_prepared, as a handle.  The C ABI owns the vector and the consume-once rule; this
    object only says that (and how long) it is there."""

    def __init__(self, which, part, n):
        self.which, self.part, self.n = which, part, n

    def __len__(self):
        return self.n

    def to_host(self, eng):
        return eng.prepared_download(self.which, self.part)


class FlasheCipher(object):
    """Drop-in for federatedml.secureprotol.jzf_flashe.FlasheCipher (jzf_flashe.py:228-666)."""

    _engine_cls = Engine      # the device context type (the HIP engine; tests may inject a double)

    def __init__(self, int_bits, mask="double", device=0):
        if 128 // int_bits < 1:          # the reference divides by merge_size = 128 // int_bits
            raise ZeroDivisionError("integer division or modulo by zero")
        self.int_bits = int_bits
        self.masking_scheme = mask
        # the same public attributes as jzf_flashe.py:230-260 (callers read and assign several of them)
        for name in ("uuid", "exchanged_keys", "masks", "total", "prp_seed", "guest_uuid", "idx",
                     "index_prefix_for_add", "index_prefix_for_minus", "iter_index_bytes", "num_clients", "num_params"):
            setattr(self, name, None)
        self.prp_seed_len = 256
        self.iter_index = -1
        self.encrypt_base = self.decrypt_base = 0
        # mask caches: filled by prepare_*(), consumed (deleted) by the next encrypt / decrypt
        self.next_iter_encrypt_prepared = {}
        self.next_iter_decrypt_prepared = {}
        self.next_iter_decrypt_prepared_idx = {}

        self._device = device
        self._engine = None
        self._key = None
        self._ctx_holds = 0           # which precompute caches the engine's ctx may hold (PREPARED_ENCRYPT | PREPARED_DECRYPT bits)
        self._pt_stage = None         # this cipher's upload block for host plaintexts of device-resident encrypts (_staged)

    # ------------------------------------------------------------------ simple setters
    def set_num_clients(self, num_clients):
        self.num_clients = num_clients

    def set_self_uuid(self, uuid):
        self.uuid = uuid

    def set_exchanged_keys(self, exchanged_keys):     # jzf_flashe.py:268-275
        self.exchanged_keys = exchanged_keys
        for k, v in exchanged_keys.items():
            if k == self.uuid:
                self.idx = v[0]
            elif v[2] == "guest":
                self.guest_uuid = k

    def get_guest_uuid(self):
        return self.guest_uuid

    def generate_prp_seed(self, assigned_seed=None):  # jzf_flashe.py:280-295
        if assigned_seed is None:
            seed = os.urandom(self.prp_seed_len // BITS_PER_BYTES)            # 32 random bytes
        else:
            # the reference masks to prp_seed_len BITS but serialises to prp_seed_len BYTES (:286-292)
            as_int = assigned_seed if isinstance(assigned_seed, int) else int.from_bytes(assigned_seed, 'big')
            seed = (as_int & ((1 << self.prp_seed_len) - 1)).to_bytes(self.prp_seed_len, 'big')
        self.prp_seed = seed
        # AESCipher.generate_key (jzf_aes.py:21-28): the AES-256 key is the low 256 bits, big-endian
        self._key = (int.from_bytes(seed, 'big') & (256 ** 32 - 1)).to_bytes(32, 'big')
        if self._engine is None:
            self._engine = self._engine_cls(self._key, self.int_bits, device=self._device)
        else:
            self._engine.set_key(self._key)

    def get_prp_seed(self):
        return self.prp_seed

    def set_iter_index(self, iter_index):             # jzf_flashe.py:300-304
        self.encrypt_base = 0
        self.decrypt_base = 0
        self.iter_index = iter_index
        self.iter_index_bytes = iter_index.to_bytes(4, 'big')

    def get_idx_list(self):
        return [self.idx]

    def set_num_params(self, num_params):
        self.num_params = num_params

    @property
    def engine(self):
        """The device context (None until a PRP seed is set)."""
        return self._engine

    # ------------------------------------------------------------------ prefix selection
    @staticmethod
    def _idx_of(prefix):
        return int.from_bytes(prefix[4:8], 'big')

    def set_idx_list_single(self, raw_idx_list=None, mode="encrypt"):   # jzf_flashe.py:306-343
        if mode == "encrypt":
            self.index_prefix_for_add = self.iter_index_bytes + self.idx.to_bytes(4, 'big')
        else:
            if self.masks is None:
                self.index_prefix_for_minus = [self.iter_index_bytes + idx.to_bytes(4, 'big') for idx in raw_idx_list]
            else:
                # dense minus-mask: client c's COMPACT stream scattered to its location list
                eng = self._engine
                locs = [np.asarray(m, dtype=np.uint32) for m in self.masks]
                for c, l in enumerate(locs):
                    if len(l) and int(l.max()) >= self.total:
                        raise IndexError(f"index {int(l.max())} is out of bounds for axis 0 with size {self.total}")
                dloc = [eng.upload(l) if len(l) else eng.alloc(16) for l in locs]
                strictly_up = all(len(l) < 2 or bool(np.all(l[1:] > l[:-1])) for l in locs)        # (what Client.sparsify emits)
                self.next_iter_decrypt_prepared["minus"] = _SparseMinus(eng, self.iter_index, dloc, [len(l) for l in locs], self.total, strictly_up)

    def set_idx_list(self, raw_idx_list=None, mode="encrypt"):          # jzf_flashe.py:345-426
        if self.masking_scheme == "single":
            return self.set_idx_list_single(raw_idx_list, mode)

        if mode == "encrypt":
            self.index_prefix_for_add = self.iter_index_bytes + self.idx.to_bytes(4, 'big')
            self.index_prefix_for_minus = self.iter_index_bytes + (self.idx + 1).to_bytes(4, 'big')
        else:
            if self.masks is None:
                raw_idx_list.sort()                                   # in place, like the reference
                run_ends, run_starts = _engine.telescope(raw_idx_list)
                # prefixes already covered by prepare_decrypt()'s masks are skipped (:372-386)
                have_add = self.next_iter_decrypt_prepared_idx.get('add', ())
                have_minus = self.next_iter_decrypt_prepared_idx.get('minus', ())
                it = self.iter_index_bytes
                self.index_prefix_for_add = [it + i.to_bytes(4, 'big') for i in run_ends if i not in have_add]
                self.index_prefix_for_minus = [it + i.to_bytes(4, 'big') for i in run_starts if i not in have_minus]
            else:
                # sparse + double: per-client run analysis on one-hot location vectors (:388-407),
                # masks indexed by DENSE position (_static_prepare_decrypt_spar as one chunk; the
                # reference's per-chunk slicing of the client list (:413-414) is a defect that
                # crashes for N_JOBS > 1 -- see DESIGN.md)
                eng = self._engine
                total = self.total
                # the run analysis works on SETS of positions (one-hot vectors in the reference): sorted, duplicates dropped
                locs = [np.unique(np.asarray(m, dtype=np.int64)) for m in self.masks]
                for l in locs:
                    if len(l) and (int(l[-1]) >= total or int(l[0]) < -total):
                        raise IndexError(f"index {int(l[-1]) if int(l[-1]) >= total else int(l[0])} is out of bounds for axis 0 with size {total}")
                locs = [np.unique(np.where(l < 0, l + total, l)).astype(np.uint32) for l in locs]         # (negative indices wrap, as in NumPy)
                va, vm = _DevVec(eng, total), _DevVec(eng, total)
                dloc = [eng.upload(l) for l in locs]
                # every list entry looks its position up in the neighbouring clients' lists on the device; no one-hot vectors
                eng.sparse_double_masks_dev(self.iter_index, dloc, [len(l) for l in locs], total, va.buf, vm.buf)
                eng.sync()
                self.next_iter_decrypt_prepared['add'] = va
                self.next_iter_decrypt_prepared['minus'] = vm

    # ------------------------------------------------------------------ encrypt
    def _check_prepared_len(self, vec, n):
        if len(vec) != n:
            raise ValueError(f"operands could not be broadcast together with shapes ({n},) ({len(vec)},) ")

    def _reconcile_prepared(self):
        """The dict entries are handles of masks the ctx holds.  A caller may drop or replace them like any attribute (the reference's
        are plain dicts: `cipher.next_iter_encrypt_prepared = {}`); whenever a handle is gone while the ctx still holds its masks, the
        ctx cache is discarded (and its HBM released) so that the two states cannot disagree."""
        eng = self._engine
        if not self._ctx_holds or eng is None:
            return
        for which, d in ((eng.PREPARED_ENCRYPT, self.next_iter_encrypt_prepared), (eng.PREPARED_DECRYPT, self.next_iter_decrypt_prepared)):
            if self._ctx_holds & which:
                h = d.get('add') if isinstance(d, dict) else None
                if not (isinstance(h, _CtxMask) and h.which == which):
                    eng.prepared_discard(which)
                    self._ctx_holds &= ~which

    # ---- device-resident operands (new) ----
    @staticmethod
    def _wants_device(value, device):
        return isinstance(value, DeviceVector) if device is None else bool(device)

    # ---- compact layout (new): int_bits <= 32 keeps device-resident vectors as uint32 arrays ----
    def _compact_ok(self):
        """The shipped un-batched jobs run int_bits = 20 (examples/configs/cnn_flashe_q16_b1_pad: quantize.int_bits): one element per
        uint64 limb moves 8 bytes for 20 useful bits, and the kernels of those widths are bound by exactly those bytes.  With
        int_bits <= 32 every DeviceVector this class produces is a uint32 array (flashe_encrypt_batch_u32_dev,
        flashe_aggregate_elem_u32_dev, flashe_aggregate_decrypt_u32_dev), and np.uint32 arrays are accepted and returned as such."""
        eng = self._engine
        if self.int_bits > 32 or not hasattr(eng, "encrypt_batch_u32_dev"):
            return False
        ask = getattr(eng, "compact_supported", None)          # the ctx knows its PRF backend and whether the chained kernels are on
        return bool(ask()) if ask is not None else os.environ.get("FLASHE_CHAIN", "1") != "0"

    def _is_u32(self, value):
        """np.uint32 arrays keep their dtype only where a uint32 holds a whole element (int_bits <= 32); at wider moduli they are
        integer arrays like any other (object-int path: full-width ciphertexts, L-limb operands for the decrypt)."""
        return self.int_bits <= 32 and isinstance(value, np.ndarray) and value.dtype == np.uint32

    def _staged(self, arr, n, limbs, elem_bytes=8):
        """A host plaintext in this cipher's own upload block, as a DeviceVector VIEW of it (round 6).

        Why not a fresh block per call: the caching allocator hands a parked block out again only after a DEVICE-WIDE synchronisation
        (csrc/blockpool.h -- the guarantee hipFree gives), and the plaintext block of the previous call was parked a moment ago: every
        encrypt(host, device=True) after the first waited there for whatever ran on ANY stream, i.e. for the previous client's encrypt
        on its own cipher's stream -- ten clients' uploads and encrypts ran strictly one after the other.  The cipher's own block is
        reused in stream order (the upload is on the ctx stream, behind the kernel that read the block last), no allocator, no device
        synchronisation: client c + 1's upload runs beside client c's encrypt.  The block (the size of one plaintext) stays with the
        cipher until release_device_buffers() or the cipher goes away; the view is only ever an INPUT of the call that made it."""
        eng = self._engine
        need = max(int(arr.nbytes), 16)
        st = self._pt_stage
        if st is None or st.nbytes < need or st.nbytes > 2 * need + (1 << 20):
            st = self._pt_stage = eng.alloc(need)
        st.upload(arr)
        return DeviceVector(eng, n, limbs, buf=st, elem_bytes=elem_bytes)

    def release_device_buffers(self):
        """Give the cipher's upload block back to the device pool (new; the reference has no device state)."""
        self._pt_stage = None

    def _on_device(self, value, full_width=False, stage=False):
        """(DeviceVector on this cipher's engine, kind of the host form).  full_width: the operand must have L limbs.  stage: a host
        array goes into the cipher's own upload block (the single operand of an encrypt) instead of a block of its own."""
        eng = self._engine
        if isinstance(value, DeviceVector):
            if value.device != getattr(eng, "device", 0):
                raise ValueError(f"DeviceVector lives on device {value.device}, this cipher on device {eng.device}")
            if value.limbs != eng.limbs and (full_width or value.limbs != 1):
                raise ValueError(f"expected {eng.limbs} limbs per element, got {value.limbs}")
            value.wait_on(eng)
            return value, ("u32" if getattr(value, "compact", False) else "u64_2d")
        if self._is_u32(value):
            if value.ndim != 1:
                raise ValueError(f"uint32 input must be [n], got {value.shape}")
            if self._compact_ok():
                if stage and hasattr(eng, "alloc"):
                    value = np.ascontiguousarray(value)
                    return self._staged(value, value.shape[0], 1, elem_bytes=4), "u32"
                return DeviceVector.from_host(eng, value), "u32"
            return DeviceVector.from_host(eng, value.astype(np.uint64)), "u32"
        limbs, kind = _to_limbs(value, eng.limbs)
        if full_width and limbs.shape[1] != eng.limbs:
            limbs = np.concatenate([limbs, np.zeros((limbs.shape[0], eng.limbs - limbs.shape[1]), dtype=np.uint64)], axis=1)
        if stage and hasattr(eng, "alloc"):
            limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
            return self._staged(limbs, limbs.shape[0], limbs.shape[1]), kind
        return DeviceVector.from_host(eng, limbs), kind

    @staticmethod
    def _deliver(out, kind, want_device):
        if want_device:
            return out.mark_ready()
        host = out.to_host()
        if getattr(out, "compact", False):                                # uint32 [n] from the device
            return host if kind == "u32" else _from_limbs(host.astype(np.uint64).reshape(-1, 1), kind)
        if kind == "u32":
            return host[:, 0].astype(np.uint32)
        return _from_limbs(host, kind)

    def _as_compact(self, dv):
        return dv if dv.compact else dv.narrowed(self._engine)

    def _as_wide(self, dv):
        return dv.widened(self._engine) if getattr(dv, "compact", False) else dv

    def _encrypt_single(self, value, device=None):                       # jzf_flashe.py:431-454
        eng = self._engine
        if self._wants_device(value, device) or isinstance(value, DeviceVector) or self._is_u32(value):
            dv, kind = self._on_device(value, stage=True)
            want_dev = self._wants_device(value, device)
            if self._compact_ok() and (dv.compact or want_dev):
                dv = self._as_compact(dv)
                out = DeviceVector(eng, len(dv), 1, elem_bytes=4)
                eng.encrypt_batch_u32_dev(self.iter_index, [self._idx_of(self.index_prefix_for_add)], SCHEME_SINGLE, len(dv), N_JOBS, [dv.buf], [out.buf])
            else:
                dv = self._as_wide(dv)
                out = DeviceVector(eng, len(dv))
                eng.encrypt_dev(self.iter_index, self._idx_of(self.index_prefix_for_add), SCHEME_SINGLE, len(dv), N_JOBS, dv.buf, dv.limbs, out.buf)
            ct = self._deliver(out, kind, want_dev)
        else:
            limbs, kind = _to_limbs(value, eng.limbs)
            ct = _from_limbs(eng.encrypt(self.iter_index, self._idx_of(self.index_prefix_for_add), SCHEME_SINGLE, N_JOBS, limbs), kind)
        if 'add' in self.next_iter_encrypt_prepared:
            del self.next_iter_encrypt_prepared['add']
        return ct

    def _encrypt_double(self, value, device=None):                       # jzf_flashe.py:456-488
        eng = self._engine
        want_dev = self._wants_device(value, device)
        prepared = 'add' in self.next_iter_encrypt_prepared
        if want_dev or prepared or isinstance(value, DeviceVector) or self._is_u32(value):
            dv, kind = self._on_device(value, stage=True)
            n = len(dv)
            if not prepared and self._compact_ok() and (dv.compact or want_dev):
                dv = self._as_compact(dv)
                out = DeviceVector(eng, n, 1, elem_bytes=4)
                eng.encrypt_batch_u32_dev(self.iter_index, [self._idx_of(self.index_prefix_for_add)], SCHEME_DOUBLE, n, N_JOBS, [dv.buf], [out.buf])
                ct = self._deliver(out, kind, want_dev)
                return ct
            dv = self._as_wide(dv)
            out = DeviceVector(eng, n)
            if not prepared:
                eng.encrypt_dev(self.iter_index, self._idx_of(self.index_prefix_for_add), SCHEME_DOUBLE, n, N_JOBS, dv.buf, dv.limbs, out.buf)
            else:
                add = self.next_iter_encrypt_prepared['add']
                minus = self.next_iter_encrypt_prepared['minus']
                self._check_prepared_len(add, n)
                if isinstance(add, _CtxMask):
                    eng.encrypt_prepared_dev(n, dv.buf, dv.limbs, out.buf)          # the ctx adds its cached masks and drops them
                    self._ctx_holds &= ~eng.PREPARED_ENCRYPT
                else:
                    eng.combine_dev(n, dv.buf, dv.limbs, add.buf, minus.buf, out.buf)
            ct = self._deliver(out, kind, want_dev)
        else:
            limbs, kind = _to_limbs(value, eng.limbs)
            ct = _from_limbs(eng.encrypt(self.iter_index, self._idx_of(self.index_prefix_for_add), SCHEME_DOUBLE, N_JOBS, limbs), kind)
        if 'add' in self.next_iter_encrypt_prepared:
            del self.next_iter_encrypt_prepared['add']
        if 'minus' in self.next_iter_encrypt_prepared:
            del self.next_iter_encrypt_prepared['minus']
        return ct

    def _begin(self, name):
        """Phase markers with the reference's own wording (jzf_aggregator.py:729,747,821-826,881-890 bracket their phases with
        LOGGER.info("begin encryption") / ("end encryption"), which its log post-processing turns into per-phase times): the same
        two INFO lines on this module's logger, the end line also carrying n and the wall time."""
        if LOGGER.isEnabledFor(logging.INFO):
            LOGGER.info("begin %s", name)
        return time.perf_counter()

    def _phase(self, name, n, t0):
        if LOGGER.isEnabledFor(logging.INFO):
            LOGGER.info("end %s (scheme=%s iter=%s n=%d seconds=%.6f)", name, self.masking_scheme, self.iter_index, n, time.perf_counter() - t0)

    def encrypt(self, plaintext, device=None):                           # jzf_flashe.py:490-504
        """device: None = the result has the form of the input (ndarray -> ndarray, DeviceVector -> DeviceVector);
        True = keep the ciphertext in HBM and return a DeviceVector; False = return a host array."""
        if self.prp_seed is not None:
            if self.masking_scheme == "double":
                self.set_idx_list(mode="encrypt")
            else:
                self.set_idx_list_single(mode="encrypt")
            if not isinstance(plaintext, (np.ndarray, DeviceVector)):
                return None
            self._reconcile_prepared()
            t0 = self._begin("encryption")
            out = self._encrypt_double(plaintext, device) if self.masking_scheme == "double" else self._encrypt_single(plaintext, device)
            self._phase("encryption", len(plaintext), t0)
            return out
        return None

    # ------------------------------------------------------------------ decrypt
    def _decrypt_single(self, value, device=None):                       # jzf_flashe.py:506-535
        eng = self._engine
        want_dev = self._wants_device(value, device)
        if want_dev or self.masks is not None or isinstance(value, DeviceVector) or self._is_u32(value):
            dv, kind = self._on_device(value, full_width=True)
            was_compact = dv.compact
            dv = self._as_wide(dv)             # (prefix LISTS and dense masks run in the one-limb layout; the result goes back compact)
            n = len(dv)
            out = DeviceVector(eng, n)
            if self.masks is None:
                minus_idx = [self._idx_of(p) for p in self.index_prefix_for_minus]
                eng.decrypt_dev(self.iter_index, [], minus_idx, n, N_JOBS, dv.buf, out.buf)
            else:
                minus = self.next_iter_decrypt_prepared['minus']
                self._check_prepared_len(minus, n)
                if isinstance(minus, _SparseMinus) and not minus.materialized and minus.it == self.iter_index:
                    eng.sparse_decrypt_dev(minus.it, minus.dloc, minus.ks, n, N_JOBS, dv.buf, out.buf, sorted_lists=minus.sorted_lists)
                else:
                    eng.combine_dev(n, dv.buf, eng.limbs, None, minus.buf, out.buf)
            if was_compact and self._compact_ok():
                out = out.mark_ready().narrowed(eng)
            res = self._deliver(out, kind, want_dev)
        else:
            limbs, kind = _to_limbs(value, eng.limbs)
            n = limbs.shape[0]
            if limbs.shape[1] != eng.limbs:
                limbs = np.concatenate([limbs, np.zeros((n, eng.limbs - limbs.shape[1]), dtype=np.uint64)], axis=1)
            minus_idx = [self._idx_of(p) for p in self.index_prefix_for_minus]
            res = _from_limbs(eng.decrypt(self.iter_index, [], minus_idx, N_JOBS, limbs), kind)
        if 'minus' in self.next_iter_decrypt_prepared:
            del self.next_iter_decrypt_prepared['minus']
        return res

    def _decrypt_double(self, value, device=None):                       # jzf_flashe.py:537-582
        eng = self._engine
        want_dev = self._wants_device(value, device)
        add_idx, minus_idx = [], []
        if self.masks is None and (self.index_prefix_for_minus or self.index_prefix_for_add):
            add_idx = [self._idx_of(p) for p in self.index_prefix_for_add]
            minus_idx = [self._idx_of(p) for p in self.index_prefix_for_minus]
        online = bool(add_idx or minus_idx)
        prepared = 'add' in self.next_iter_decrypt_prepared
        if not prepared and not online:
            _to_limbs(value, eng.limbs) if isinstance(value, np.ndarray) else None      # (conversion errors come first, as in the reference)
            raise KeyError('add')                                         # what the reference raises (:570)
        if want_dev or prepared or isinstance(value, DeviceVector) or self._is_u32(value):
            dv, kind = self._on_device(value, full_width=True)
            n = len(dv)
            if dv.compact and not prepared and self._compact_ok() and len(add_idx) == 1 and len(minus_idx) <= 1:
                # the no-dropout decrypt (and every single telescoped run) on the uint32 vector itself: a one-operand "reduce + decrypt"
                out = DeviceVector(eng, n, 1, elem_bytes=4)
                eng.aggregate_decrypt_u32_dev(self.iter_index, add_idx, minus_idx, n, N_JOBS, 0, n, [dv.buf], None, out.buf, out_elem_bytes=4)
                res = self._deliver(out, kind, want_dev)
                for d in (self.next_iter_decrypt_prepared, self.next_iter_decrypt_prepared_idx):
                    for k in ('add', 'minus'):
                        if k in d:
                            del d[k]
                return res
            was_compact = dv.compact
            dv = self._as_wide(dv)
            out = DeviceVector(eng, n)
            if not prepared:
                eng.decrypt_dev(self.iter_index, add_idx, minus_idx, n, N_JOBS, dv.buf, out.buf)
            else:
                padd = self.next_iter_decrypt_prepared['add']
                pminus = self.next_iter_decrypt_prepared['minus']
                self._check_prepared_len(padd, n)
                if isinstance(padd, _CtxMask):
                    # the ctx's cached masks, the extras (dropouts) computed online and merged in (:557-564), the cache dropped
                    eng.decrypt_prepared_dev(self.iter_index, add_idx, minus_idx, n, N_JOBS, dv.buf, out.buf)
                    self._ctx_holds &= ~eng.PREPARED_DECRYPT
                else:
                    eng.combine_dev(n, dv.buf, eng.limbs, padd.buf, pminus.buf, out.buf)
                    if online:                                            # extras merged in (:557-564)
                        eng.decrypt_dev(self.iter_index, add_idx, minus_idx, n, N_JOBS, out.buf, out.buf)
            if was_compact and self._compact_ok():
                out = out.mark_ready().narrowed(eng)
            res = self._deliver(out, kind, want_dev)
        else:
            limbs, kind = _to_limbs(value, eng.limbs)
            n = limbs.shape[0]
            if limbs.shape[1] != eng.limbs:
                limbs = np.concatenate([limbs, np.zeros((n, eng.limbs - limbs.shape[1]), dtype=np.uint64)], axis=1)
            res = _from_limbs(eng.decrypt(self.iter_index, add_idx, minus_idx, N_JOBS, limbs), kind)
        for d in (self.next_iter_decrypt_prepared, self.next_iter_decrypt_prepared_idx):
            for k in ('add', 'minus'):
                if k in d:
                    del d[k]
        return res

    def decrypt(self, ciphertext, device=None):                          # jzf_flashe.py:584-594
        """device: as for encrypt()."""
        if self.prp_seed is not None:
            if not isinstance(ciphertext, (np.ndarray, DeviceVector)):
                return None
            self._reconcile_prepared()
            t0 = self._begin("decryption")
            out = self._decrypt_double(ciphertext, device) if self.masking_scheme == "double" else self._decrypt_single(ciphertext, device)
            self._phase("decryption", len(ciphertext), t0)
            return out
        return None

    # ------------------------------------------------------------------ mask precompute
    # The masks live inside the engine's ctx (flashe_prepare_encrypt / flashe_prepare_decrypt): both streams in one launch, consumed by
    # the next flashe_encrypt_prepared_dev / flashe_decrypt_prepared_dev -- the state machine of :483-486 / :573-580 is the C ABI's, the
    # dict entries below are its handles (so callers that test `'add' in cipher.next_iter_encrypt_prepared` keep working).
    def prepare_encrypt(self):                                           # jzf_flashe.py:599-631
        (self.iter_index + 1).to_bytes(4, 'big')                          # same range check as the reference
        eng, n = self._engine, self.num_params
        eng.prepare_encrypt(self.iter_index + 1, self.idx, SCHEME_DOUBLE, n, N_JOBS)
        eng.sync()
        self.next_iter_encrypt_prepared = {'add': _CtxMask(eng.PREPARED_ENCRYPT, 'add', n), 'minus': _CtxMask(eng.PREPARED_ENCRYPT, 'minus', n)}
        self._ctx_holds |= eng.PREPARED_ENCRYPT

    def prepare_decrypt(self):                                           # jzf_flashe.py:633-666
        eng, n = self._engine, self.num_params
        eng.prepare_decrypt(self.iter_index, self.num_clients, n, N_JOBS)
        eng.sync()
        self.next_iter_decrypt_prepared['add'] = _CtxMask(eng.PREPARED_DECRYPT, 'add', n)
        self.next_iter_decrypt_prepared['minus'] = _CtxMask(eng.PREPARED_DECRYPT, 'minus', n)
        self._ctx_holds |= eng.PREPARED_DECRYPT
        self.next_iter_decrypt_prepared_idx['add'] = [self.num_clients]
        self.next_iter_decrypt_prepared_idx['minus'] = [0]

    # ------------------------------------------------------------------ arbiter reduce (new)
    def aggregate(self, ciphertexts, packed=False, device=None):
        """Server-side reduce of a list of ciphertext vectors.  The reference has no such method;
        this equals Arbiter.aggregate_model's flashe branch: element-wise
        (jzf_aggregator.py:424-430) or, with packed=True, on the bit-packed integers
        (jzf_aggregator.py:406-419), returned unpacked.  Operands may be DeviceVectors (they stay where they are);
        device: as for encrypt() -- None returns a DeviceVector iff the first operand is one."""
        eng = self._engine or self._engine_cls(bytes(32), self.int_bits, device=self._device)
        t0 = self._begin("aggregation")
        out = aggregate(ciphertexts, self.int_bits, packed=packed, device=self._device, _engine=eng, keep_on_device=device)
        self._phase("aggregation", len(out), t0)
        return out


def aggregate(ciphertexts, int_bits, packed=False, device=0, _engine=None, keep_on_device=None):
    """reduce(lambda x, y: (x + y) % mod, models) on the GPU.

    packed=False: mod = 1 << int_bits per element (jzf_aggregator.py:424-430).
    packed=True : every operand is bit-packed (JZFTransferableWeights.compress,
    jzf_weights.py:155-195), added mod 1 << (int_bits * n) with carries crossing element
    boundaries (jzf_aggregator.py:406-419), then unpacked again (decompress, :197-231).
    keep_on_device: True = return a DeviceVector, False = a host array, None = what the first operand is."""
    if len(ciphertexts) == 0:
        raise TypeError("reduce() of empty sequence with no initial value")
    eng = _engine or Engine(bytes(32), int_bits, device=device)
    want_dev = isinstance(ciphertexts[0], DeviceVector) if keep_on_device is None else bool(keep_on_device)
    is_compact = lambda c: (isinstance(c, DeviceVector) and c.compact) or (isinstance(c, np.ndarray) and c.dtype == np.uint32)    # noqa: E731
    if (int_bits <= 32 and not packed and hasattr(eng, "aggregate_elem_u32_dev") and all(is_compact(c) for c in ciphertexts)):
        # compact layout: uint32 operands (handles or arrays), uint32 result -- half the bytes of the one-limb reduce
        ops, held = [], {}
        for c in ciphertexts:
            if id(c) not in held:
                if isinstance(c, DeviceVector):
                    c.wait_on(eng)
                    held[id(c)] = c
                else:
                    held[id(c)] = DeviceVector.from_host(eng, c.reshape(-1))
            ops.append(held[id(c)])
        n = len(ops[0])
        if any(len(o) != n for o in ops):
            raise ValueError("operands could not be broadcast together")
        out = DeviceVector(eng, n, 1, elem_bytes=4)
        eng.aggregate_elem_u32_dev([o.buf for o in ops], n, out.buf)
        return out.mark_ready() if want_dev else out.to_host()
    # (a compact operand among one-limb ones, or the packed reduce: widened on the device)
    ciphertexts = [c.widened(eng) if (isinstance(c, DeviceVector) and c.compact) else (c.astype(np.uint64) if (isinstance(c, np.ndarray) and c.dtype == np.uint32) else c)
                   for c in ciphertexts]
    # an operand passed several times (the notebook's `[ct] * num_clients`) is converted and uploaded once
    seen, conv = {}, []
    for c in ciphertexts:
        if id(c) not in seen:
            if isinstance(c, DeviceVector):
                if c.limbs != eng.limbs:
                    raise ValueError(f"expected {eng.limbs} limbs per element, got {c.limbs}")
                c.wait_on(eng)
                seen[id(c)] = (c, "u64_2d")
            else:
                a, k = _to_limbs(np.asarray(c) if not isinstance(c, np.ndarray) else c, eng.limbs)
                if a.shape[1] != eng.limbs:
                    a = np.concatenate([a, np.zeros((a.shape[0], eng.limbs - a.shape[1]), dtype=np.uint64)], axis=1)
                seen[id(c)] = (a, k)
        conv.append(seen[id(c)])
    kind = conv[0][1]
    arrs = [a for a, _k in conv]
    n = len(arrs[0])
    if any(len(a) != n for a in arrs):
        raise ValueError("operands could not be broadcast together")
    if (not packed and not want_dev and not any(isinstance(a, DeviceVector) for a in arrs) and hasattr(eng, "aggregate_elem")
            and len(arrs) <= 64 and len({id(a) for a in arrs}) == len(arrs)):
        # distinct operands, all on the host, result wanted on the host: the host-pointer twin (chunk-pipelined upload / reduce /
        # download); repeated operands take the path below, which uploads each array once
        return _from_limbs(eng.aggregate_elem(arrs), kind)
    dev = {}
    for a in arrs:
        if id(a) not in dev:
            dev[id(a)] = a.buf if isinstance(a, DeviceVector) else eng.upload(a)
    dsrc = [dev[id(a)] for a in arrs]
    out = DeviceVector(eng, n)
    if not packed:
        eng.aggregate_elem_dev(dsrc, n, out.buf)
    else:
        total_bits = n * int_bits
        n_limbs = (total_bits + 63) // 64
        dpk = [eng.alloc(max(n_limbs * 8, 16)) for _ in arrs]
        for s_, p_ in zip(dsrc, dpk):
            eng.pack_dev(n, s_, p_)
        dsum = eng.alloc(max(n_limbs * 8, 16))
        eng.aggregate_packed_dev(dpk, n_limbs, total_bits, dsum)
        eng.unpack_dev(n, dsum, out.buf)
    if want_dev:
        return out.mark_ready()
    return _from_limbs(out.to_host(), kind)
