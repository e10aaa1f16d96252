"""Quantise / batch codec either side of the cipher, on the GPU -- mirrors what the FLASHE path uses from
federatedml/secureprotol/jzf_quantize.py and jzf_aciq.py: the four static codec functions (:55-67, :102-107,
:162-185, :234-251), the ACIQ clipping threshold (jzf_aciq.py:10-27) and the transport-free part of
QuantizingClient (quantize / unquantize / normalize / unnormalize, jzf_quantize.py:394-564): same names,
arguments and results.

Stochastic rounding: the reference draws its uniforms with ``np.random.random(size)`` (numpy's
global legacy generator).  The mirror consumes the *same* stream, so with the same seed the quantised
integers are bit-identical: small layers draw on the host and ship the numbers, layers of at least
DEVICE_RNG_MIN elements generate them ON THE DEVICE from NumPy's own MT19937 state
(`flashe_mt19937_random_dev`: the state is read, advanced on the device exactly as NumPy would advance
it, and put back -- host draws before and after continue one stream; FLASHE_DEVICE_RNG=0 turns it
off).  Pass ``uniforms=`` to supply the draws yourself.
"""
import os

import numpy as np

DEVICE_RNG_MIN = 1 << 16

from .engine import Engine

_engines = {}


def _engine(int_bits=64, device=0):
    key = (int_bits, device)
    if key not in _engines:
        _engines[key] = Engine(bytes(32), int_bits, device=device)      # the codec needs no PRF key
    return _engines[key]


def _as_object(arr):
    return np.asarray(arr, dtype=np.uint64).astype(object)


def _static_quantize_padding_asymmetric(value, alpha, int_bits, uniforms=None, device=0, as_object=True):
    """jzf_quantize.py:55-67."""
    value = np.asarray(value)
    shape = value.shape
    eng = _engine(64, device)
    n = int(value.size)
    if (uniforms is None and n >= DEVICE_RNG_MIN and os.environ.get("FLASHE_DEVICE_RNG", "1") != "0"
            and hasattr(eng, "numpy_random_dev") and np.random.get_state()[0] == "MT19937"):
        # np.random.random(shape) generated where it is consumed: no host loop over n draws, no 8 B / element upload
        x = np.ascontiguousarray(value).reshape(-1)
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        dx, dq = eng.upload(x), eng.alloc(8 * n)
        du = eng.numpy_random_dev(n)
        eng.quantize_dev(n, dx, x.dtype == np.float64, alpha, int_bits, du, dq)
        q = dq.download(np.uint64, n).reshape(shape)
    else:
        u = np.random.random(shape) if uniforms is None else np.asarray(uniforms, dtype=np.float64)
        q = eng.quantize(value.reshape(-1), alpha, int_bits, u.reshape(-1)).reshape(shape)
    return _as_object(q) if as_object else q


def _static_unquantize_padding_asymmetric(value, alpha, int_bits, num_clients, device=0):
    """jzf_quantize.py:102-107.  value: object ints / uint64 [n] / uint64 limbs [n, 2]."""
    value = np.asarray(value)
    if value.dtype == object:
        flat = value.reshape(-1)
        limbs = np.empty((flat.shape[0], 2), dtype=np.uint64)
        m64 = (1 << 64) - 1
        limbs[:, 0] = (flat & m64).astype(np.uint64)
        limbs[:, 1] = ((flat >> 64) & m64).astype(np.uint64)
        out = _engine(128, device).unquantize(limbs, alpha, int_bits, num_clients)
        return out.reshape(value.shape)
    if value.ndim == 2 and value.shape[1] == 2:
        return _engine(128, device).unquantize(value, alpha, int_bits, num_clients)
    return _engine(64, device).unquantize(value.reshape(-1), alpha, int_bits, num_clients).reshape(value.shape)


def _static_batching_padding_asymmetric(array, int_bits, element_bits, factor, device=0, as_object=True):
    """jzf_quantize.py:162-185: returns the batched int_bits-wide integers."""
    eng = _engine(int_bits, device)
    vals = np.asarray(array)
    if vals.dtype == object:
        vals = vals.astype(np.uint64)
    out = eng.batch(vals.reshape(-1), element_bits + factor)
    if not as_object:
        return out
    ints = out[:, 0].astype(object)
    if out.shape[1] == 2:
        ints = ints | (out[:, 1].astype(object) << 64)
    return ints


def _static_unbatching_padding_asymmetric(array, int_bits, element_bits, factor, device=0):
    """jzf_quantize.py:234-251: returns n_batches * batch_size values (zero padding included)."""
    eng = _engine(int_bits, device)
    arr = np.asarray(array)
    if arr.dtype == object:
        flat = arr.reshape(-1)
        limbs = np.empty((flat.shape[0], eng.limbs), dtype=np.uint64)
        m64 = (1 << 64) - 1
        limbs[:, 0] = (flat & m64).astype(np.uint64)
        if eng.limbs == 2:
            limbs[:, 1] = ((flat >> 64) & m64).astype(np.uint64)
        arr = limbs
    return eng.unbatch(arr, element_bits + factor)


# ------------------------------------------------------------------------------------------------------------------
# ACIQ clipping threshold and the client-side orchestration
# ------------------------------------------------------------------------------------------------------------------
class ACIQ(object):
    """jzf_aciq.py:4-27: alpha = (optimal clipping multiple for `num_bits`-bit quantisation of a Gaussian) x sigma.  The table is the
    published ACIQ one as the reference carries it (indices 2..31; more than 31 bits use the last entry)."""

    _ALPHA_GAUS = [None, None, 1.710635, 2.151593, 2.559136, 2.936201, 3.286914, 3.615114,
                   3.924035, 4.216331, 4.494167, 4.759313, 5.013188, 5.257151, 5.491852, 5.719160,
                   5.938345, 6.150141, 6.356593, 6.560495, 6.752936, 6.931921, 7.106395, 7.350340,
                   7.482915, 7.691728, 7.668494, 7.583591, 7.583591, 8.326501, 8.171210, 8.171210]

    def __init__(self, num_bits):
        self.num_bits = num_bits

    def _alpha_opt(self):
        return self._ALPHA_GAUS[31] if self.num_bits > 31 else self._ALPHA_GAUS[self.num_bits]

    def get_alpha_gaus(self, min, max, size):                       # jzf_aciq.py:10-19
        gaussian_const = (0.5 * 0.35) * (1 + (np.pi * np.log(4)) ** 0.5)
        sigma = ((max - min) * gaussian_const) / ((2 * np.log(size)) ** 0.5)
        return self._alpha_opt() * sigma

    def get_alpha_gaus_direct(self, sigma):                         # jzf_aciq.py:21-27
        return self._alpha_opt() * sigma


def _loop_dtype(arr_dtype, scalar):
    """The dtype NumPy computes `array <op> scalar` in -- asked of the running NumPy itself, so a Python float stays weak
    and a np.float64 scalar promotes a float32 array exactly as it does for the reference on this installation."""
    return (np.zeros(1, dtype=arr_dtype) + scalar).dtype


class QuantizingClient(object):
    """jzf_quantize.py:336-564 without the federation transport (secure + padding path, which is what every shipped FLASHE job
    configures): per-layer alpha from the previous global model's std, asymmetric stochastic quantisation (+ batching), and the
    normalise / unnormalise bookkeeping around it.  `weights` is anything with `.walking_order` and `._weights` (JZFOrderDictWeights'
    surface).  The arithmetic runs in the HIP kernels; uniforms come from NumPy's global generator exactly as in the reference."""

    def __init__(self, int_bits, from_arbiter=None, to_arbiter=None, batch=False, element_bits=16, padding=True, secure=True, device=0):
        # the reference's other branches: padding=False leaves `ret` unassigned in quantize / unquantize (UnboundLocalError, its
        # code there is commented out, :466-479, :518-531), secure=False takes the plain-text r_max path this engine does not mirror
        if padding is False or secure is False:
            raise NotImplementedError("QuantizingClient mirrors the secure + padding path (jzf_quantize.py:436-465, :509-517); "
                                      "padding=False / secure=False are not implemented")
        self.int_bits, self.batch, self.element_bits, self.padding, self.secure = int_bits, batch, element_bits, padding, secure
        self.from_arbiter, self.to_arbiter = from_arbiter, to_arbiter
        self.num_clients = None
        self.iter = 0
        self.r_max_list = self.alpha_list = self.shape_list = self.layer_size_list = None
        self.expected_mean_for_first_round = 0.0
        self.expected_std_for_first_round = 1.0
        self.past_layer_mean_list, self.past_layer_std_list = [], []
        self._device = device

    def set_iter(self, iter):
        self.iter = iter

    def receive_num_clients(self):
        if self.from_arbiter is not None:
            self.num_clients = self.from_arbiter.get(idx=0, suffix=(self.iter, 'num_clients'))
        return self.num_clients

    def set_layer_size_list(self, weights):                          # host, :380-392 (the guest's send_* does the same when secure)
        self.layer_size_list = [weights._weights[k].size for k in weights.walking_order]
        for _ in self.layer_size_list:
            self.past_layer_mean_list.append(self.expected_mean_for_first_round)
            self.past_layer_std_list.append(self.expected_std_for_first_round)

    send_layer_size_list = set_layer_size_list

    def quantize(self, weights):                                     # :394-491
        aciq = ACIQ(self.element_bits)
        alpha_list = []
        for i, _size in enumerate(self.layer_size_list):
            alpha = aciq.get_alpha_gaus_direct(self.past_layer_std_list[i])
            if alpha == 0:
                alpha = 0.1
            alpha_list.append(alpha)
        self.r_max_list, self.alpha_list = [], []
        if self.batch:
            self.shape_list = []
        factor = int(np.ceil(np.log2(self.num_clients)))
        layer_cnt = 0
        for k in weights.walking_order:
            if k == 'zzz':                                           # the sparsifier's trailing layer (:433-435)
                alpha = 1.0
            else:
                alpha = alpha_list[layer_cnt]
                self.r_max_list.append(alpha * self.num_clients)
                self.alpha_list.append(alpha)
            layer = np.asarray(weights._weights[k])
            shape = layer.shape
            flat = layer.flatten()
            if self.batch:
                self.shape_list.append(shape)
            want = _loop_dtype(flat.dtype, alpha)                    # the dtype `np.clip(value, -alpha, alpha) + alpha` runs in
            if flat.dtype != want:
                flat = flat.astype(want)                             # exact (float32 -> float64)
            elements = _static_quantize_padding_asymmetric(flat, float(alpha), self.element_bits, device=self._device, as_object=False)
            if self.batch:
                weights._weights[k] = _static_batching_padding_asymmetric(elements, self.int_bits, self.element_bits, factor, device=self._device)
            else:
                weights._weights[k] = _as_object(elements).reshape(shape)
            layer_cnt += 1
        return weights

    def unquantize(self, weights):                                   # :493-540
        factor = int(np.ceil(np.log2(self.num_clients)))
        for layer_cnt, k in enumerate(weights.walking_order):
            alpha = self.alpha_list[layer_cnt]
            layer = np.asarray(weights._weights[k])
            flat = layer.flatten()
            if self.batch:
                shape = self.shape_list[layer_cnt]
                size = int(np.prod(shape))
                flat = _static_unbatching_padding_asymmetric(flat, self.int_bits, self.element_bits, factor, device=self._device)[:size]
            else:
                shape = layer.shape
            ret = _static_unquantize_padding_asymmetric(flat, float(alpha), self.element_bits, self.num_clients, device=self._device)
            weights._weights[k] = np.asarray(ret, dtype=np.float64).reshape(shape)
        return weights

    def _shift(self, arr, shift):
        """arr <- arr + shift with NumPy's in-place semantics (loop dtype from the scalar, result cast back), on the device."""
        a = np.ascontiguousarray(arr)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        eng = _engine(64, self._device)
        wide = a.dtype == np.float32 and _loop_dtype(a.dtype, shift) == np.float64
        d = eng.upload(a.reshape(-1))
        eng.shift_dev(a.size, d, a.dtype == np.float64, float(shift), wide)
        return d, a

    def normalize(self, weights):                                    # :542-547
        for layer_cnt, k in enumerate(weights.walking_order):
            mean = self.past_layer_mean_list[layer_cnt]
            d, a = self._shift(weights._weights[k], -mean)
            weights._weights[k] = d.download(a.dtype, a.size).reshape(a.shape)
        return weights

    def unnormalize(self, weights):                                  # :549-564
        for layer_cnt, k in enumerate(weights.walking_order):
            d, a = self._shift(weights._weights[k], self.past_layer_mean_list[layer_cnt])
            host = d.download(a.dtype, a.size).reshape(a.shape)
            weights._weights[k] = host
            # The statistics the next round's alpha is derived from: np.mean / np.std of the array just produced, on the host, exactly
            # the calls the reference makes (:558-561) -- same summation order (NumPy's pairwise loops), same scalar type (np.float64,
            # or np.float32 for a float32 layer, which decides the dtype the next round's `-=` and clip / scale arithmetic run in).
            # A parallel device reduction (flashe_mean_std_dev) agrees only to ~1e-12, and one ulp of std moves alpha = alpha_opt * std
            # and with it stochastic-rounding results of later rounds; the layer is on the host at this point anyway.
            self.past_layer_mean_list[layer_cnt] = np.mean(host)
            self.past_layer_std_list[layer_cnt] = np.std(host)
        return weights
