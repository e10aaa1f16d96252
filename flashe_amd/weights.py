"""Wire format of the reference on top of the GPU bit-packing kernels: the integer that
`JZFTransferableWeights.compress` builds for a layer and `decompress` takes apart
(federatedml/framework/jzf_weights.py:155-231, helpers :36-137), and the location coding of sparse
uploads (`Client.sparsify`, jzf_aggregator.py:615-623: `_to_bytes(locations, total.bit_length())`).

A compressed layer is ONE Python int: sum_j x[j] << (bits * (n - 1 - j)) (element 0 most significant;
the per-chunk `_to_bytes` results merged at :180-182 give exactly this integer).  The heavy part
(field packing / unpacking) runs in `pack_kernel` / `unpack_kernel`; Python only converts between the
limb array and the int."""
import numpy as np

from .cipher import _from_limbs, _to_limbs
from .engine import Engine

_engines = {}


def _engine(bits, device=0):
    key = (bits, device)
    if key not in _engines:
        _engines[key] = Engine(bytes(32), bits, device=device)          # the codec needs no PRF key
    return _engines[key]


def to_big_int(flatten_array, num_bits, device=0):
    """`_to_bytes(flatten_array, num_bits)` / `_to_bytes_old` (jzf_weights.py:36-84) -> (int, length).  A `DeviceVector` (a ciphertext
    that stayed in HBM) is packed where it lies: only the n * num_bits / 8 packed bytes come down."""
    from .engine import DeviceVector
    eng = _engine(num_bits, device)
    if isinstance(flatten_array, DeviceVector):
        dv = flatten_array.widened(eng) if flatten_array.compact else flatten_array
        n = len(dv)
        if n == 0:
            return 0, 0
        if dv.limbs != eng.limbs:
            raise ValueError(f"expected {eng.limbs} limbs per element for {num_bits}-bit fields, got {dv.limbs}")
        dv.wait_on(eng)
        n_limbs = (n * num_bits + 63) // 64
        packed = eng.alloc(max(8 * n_limbs, 16))
        eng.pack_dev(n, dv.buf, packed)
        return int.from_bytes(packed.download(np.uint64, n_limbs).tobytes(), "little"), n
    arr = np.asarray(flatten_array)
    if arr.dtype != object and arr.dtype != np.uint64:
        arr = arr.astype(object)
    limbs, _kind = _to_limbs(arr.reshape(-1) if arr.dtype == object else arr, eng.limbs)
    n = limbs.shape[0]
    if n == 0:
        return 0, 0
    if limbs.shape[1] != eng.limbs:
        limbs = np.concatenate([limbs, np.zeros((n, eng.limbs - limbs.shape[1]), dtype=np.uint64)], axis=1)
    packed = eng.pack(limbs)
    return int.from_bytes(packed.tobytes(), "little"), n


def from_big_int(big_int, length, num_bits, device=0, as_object=True, as_device=False):
    """`_from_bytes(...)` followed by `reverse()` (jzf_weights.py:87-137, :225-228): the values in their
    original order.  as_device: unpacked in HBM and returned as a `DeviceVector` (the arbiter's operands of `aggregate`)."""
    from .engine import DeviceVector
    eng = _engine(num_bits, device)
    if length == 0:
        return DeviceVector(eng, 0) if as_device else np.array([], dtype=object if as_object else np.uint64)
    n_limbs = (length * num_bits + 63) // 64
    packed = np.frombuffer(int(big_int).to_bytes(n_limbs * 8, "little"), dtype=np.uint64)
    if as_device:
        out = DeviceVector(eng, length)
        eng.unpack_dev(length, eng.upload(packed), out.buf)
        return out.mark_ready()
    out = eng.unpack(packed, length)
    return _from_limbs(out, "object") if as_object else out


class TransferableWeights(object):
    """The compress / decompress pair of JZFTransferableWeights (jzf_weights.py:140-231) for a dict of
    integer layers; transport (segment transfer, pickling) is out of scope."""

    def __init__(self, weights, bits, need_compress=True, shape=None, device=0):
        self._bits, self._weights, self._shape, self._device = bits, weights, shape, device
        if self._bits is not None and need_compress:
            self.compress()

    def compress(self):
        res, self._shape = {}, {}
        from .engine import DeviceVector
        for k, w in self._weights.items():
            if isinstance(w, DeviceVector):                  # a ciphertext that stayed in HBM: packed there, only the packed bytes come down
                self._shape[k] = (len(w),)
                res[k] = to_big_int(w, self._bits, self._device)[0]
                continue
            w = np.asarray(w)
            self._shape[k] = w.shape
            res[k] = to_big_int(w.flatten(), self._bits, self._device)[0]
        self._weights = res

    def decompress(self, as_device=False):
        """as_device: every layer comes back as a flat `DeviceVector` in HBM (the arbiter's operands of `aggregate`) instead of an object array."""
        out = {}
        for k, big in self._weights.items():
            shape = self._shape[k]
            n = int(np.prod(shape))
            v = from_big_int(big, n, self._bits, self._device, as_device=as_device)
            out[k] = v if as_device else v.reshape(shape)
        return out

    @property
    def unboxed(self):
        return self._weights


class Sparsifier(object):
    """`Client.sparsify` (jzf_aggregator.py:578-623) on the GPU: layer-wise top-s% selection with residual
    accumulation, locations returned bit-packed as the reference sends them.

    Differences kept explicit: the ranking ignores the residual exactly like the reference (it ranks |layer|
    before `flatten += remain`); ties at the k-th magnitude go to the higher index (numpy's default argsort
    leaves them unspecified)."""

    def __init__(self, sparsity, device=0):
        self._sparsity = sparsity
        self._device = device
        self._remain_host = None            # name -> residual array (what the reference keeps in self.remain_weights)
        self._remain_dev = None             # or: (DeviceBuffer of the flat residuals, names, sizes, dtype) -- they stay in HBM between rounds
        self.shape_dict_used_for_sparsification = None

    @property
    def remain_weights(self):
        """name -> residual, as the reference's attribute.  The residuals live on the device between rounds (they are only ever read
        and updated by the next sparsify); reading this attribute downloads them."""
        if self._remain_dev is not None and self._remain_host is None:
            buf, names, sizes, dt = self._remain_dev
            flat = buf.download(dt, sum(sizes))
            self._remain_host, o = {}, 0
            for name, n in zip(names, sizes):
                self._remain_host[name] = flat[o:o + n]
                o += n
        return self._remain_host

    @remain_weights.setter
    def remain_weights(self, value):
        self._remain_host, self._remain_dev = value, None

    def sparsify(self, weights, walking_order=None):
        """weights: dict name -> float ndarray, replaced IN PLACE by the compact masked layers.
        Returns (encoded_locations, length, bits, total) like the reference."""
        eng = _engine(128, self._device)
        order = list(walking_order) if walking_order is not None else sorted(weights.keys(), key=str)
        base, locations, shapes = 0, [], {}
        layers, ks = [], []
        for k in order:
            layer = np.asarray(weights[k])
            shapes[k] = layer.shape
            layers.append(layer.reshape(-1))
            ks.append(max(1, int(np.floor(self._sparsity * int(layer.size)))))
        sizes = [int(l.size) for l in layers]
        dtypes = {l.dtype for l in layers}
        dt = next(iter(dtypes)) if len(dtypes) == 1 else None
        if hasattr(eng, "sparsify_model") and dt in (np.float32, np.float64):
            # every layer in one set of launches (a layer-by-layer walk is ~12 launches and three synchronous transfers PER LAYER), the
            # layers copied straight into one flat device buffer, the residuals kept in HBM from round to round
            dev = self._remain_dev
            if dev is None or dev[1] != order or dev[2] != sizes or dev[3] != dt:
                host = self.remain_weights or {}            # (downloads what a differently shaped earlier round left on the device)
                if not any(host.get(k) is not None for k in order):
                    # the first round: no residual yet -- zeroed where it will live, not 4 or 8 bytes per value of zeros over PCIe
                    buf = eng.alloc(max(sum(sizes) * np.dtype(dt).itemsize, 16))
                    eng.memset_dev(buf, 0, buf.nbytes)
                    dev = (buf, order, sizes, dt)
                else:
                    flat = np.concatenate([np.ascontiguousarray(host[k], dtype=dt).reshape(-1) if host.get(k) is not None else np.zeros(n, dtype=dt)
                                           for k, n in zip(order, sizes)]) if sizes else np.zeros(0, dtype=dt)
                    dev = (eng.upload(flat) if flat.size else eng.alloc(16), order, sizes, dt)
            results = eng.sparsify_model(layers, ks, dev[0], dt)
            self._remain_dev, self._remain_host = dev, None
        else:
            remain = dict(self.remain_weights or {})
            results = []
            for k, layer, k_l in zip(order, layers, ks):
                r = remain.get(k)
                if r is None:
                    r = np.zeros(layer.size, dtype=layer.dtype if layer.dtype in (np.float32, np.float64) else np.float64)
                loc, vals, remain[k] = eng.sparsify(layer, k_l, r)
                results.append((loc, vals))
            self.remain_weights = remain
        for k, layer, (loc, vals) in zip(order, layers, results):
            weights[k] = vals
            locations.append(loc.astype(np.uint64) + np.uint64(base))
            base += int(layer.size)
        if self.shape_dict_used_for_sparsification is None:
            self.shape_dict_used_for_sparsification = shapes
        all_loc = np.concatenate(locations) if locations else np.zeros(0, dtype=np.uint64)
        bits = int(base).bit_length()
        encoded, le = to_big_int(all_loc, bits, self._device)
        return encoded, le, bits, base
