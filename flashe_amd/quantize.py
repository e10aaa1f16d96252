"""Quantise / batch codec either side of the cipher, on the GPU -- mirrors the four functions the
FLASHE path uses from federatedml/secureprotol/jzf_quantize.py (QuantizingClient.quantize /
unquantize, :394-564): same names, arguments and results.

Stochastic rounding: the reference draws its uniforms with ``np.random.random(size)`` (numpy's
global legacy generator).  The mirror makes the *same* host-side draw and ships the numbers to the
kernel, so with the same seed the quantised integers are bit-identical; pass ``uniforms=`` to
supply the draws yourself.
"""
import numpy as np

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
    u = np.random.random(shape) if uniforms is None else np.asarray(uniforms, dtype=np.float64)
    q = _engine(64, device).quantize(value.reshape(-1), alpha, int_bits, u.reshape(-1)).reshape(shape)
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
