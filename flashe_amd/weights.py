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
    """`_to_bytes(flatten_array, num_bits)` / `_to_bytes_old` (jzf_weights.py:36-84) -> (int, length)."""
    eng = _engine(num_bits, device)
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


def from_big_int(big_int, length, num_bits, device=0, as_object=True):
    """`_from_bytes(...)` followed by `reverse()` (jzf_weights.py:87-137, :225-228): the values in their
    original order."""
    eng = _engine(num_bits, device)
    if length == 0:
        return np.array([], dtype=object if as_object else np.uint64)
    n_limbs = (length * num_bits + 63) // 64
    packed = np.frombuffer(int(big_int).to_bytes(n_limbs * 8, "little"), dtype=np.uint64)
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
        for k, w in self._weights.items():
            w = np.asarray(w)
            self._shape[k] = w.shape
            res[k] = to_big_int(w.flatten(), self._bits, self._device)[0]
        self._weights = res

    def decompress(self):
        out = {}
        for k, big in self._weights.items():
            shape = self._shape[k]
            n = int(np.prod(shape))
            out[k] = from_big_int(big, n, self._bits, self._device).reshape(shape)
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
        self.remain_weights = None
        self.shape_dict_used_for_sparsification = None

    def sparsify(self, weights, walking_order=None):
        """weights: dict name -> float ndarray, replaced IN PLACE by the compact masked layers.
        Returns (encoded_locations, length, bits, total) like the reference."""
        eng = _engine(128, self._device)
        if self.remain_weights is None:
            self.remain_weights = {}
        order = walking_order if walking_order is not None else sorted(weights.keys(), key=str)
        base, locations, shapes = 0, [], {}
        for k in order:
            layer = np.asarray(weights[k])
            shapes[k] = layer.shape
            size = int(np.prod(layer.shape))
            idx = max(1, int(np.floor(self._sparsity * size)))
            remain = self.remain_weights.get(k)
            if remain is None:
                remain = np.zeros(size, dtype=layer.dtype if layer.dtype in (np.float32, np.float64) else np.float64)
            loc, vals, new_remain = eng.sparsify(layer.reshape(-1), idx, remain)
            weights[k] = vals
            self.remain_weights[k] = new_remain
            locations.append(loc.astype(np.uint64) + np.uint64(base))
            base += size
        if self.shape_dict_used_for_sparsification is None:
            self.shape_dict_used_for_sparsification = shapes
        all_loc = np.concatenate(locations) if locations else np.zeros(0, dtype=np.uint64)
        bits = int(base).bit_length()
        encoded, le = to_big_int(all_loc, bits, self._device)
        return encoded, le, bits, base
