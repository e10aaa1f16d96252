"""Transport-free mirror of the adapter that drives the cipher in the reference
(federatedml/framework/homo/procedure/jzf_flashe_block.py): the `_Client` forwarders (:120-174) and
the arbiter's `dynamic_masking` cost model (:89-117).  Key exchange, uuid sync and the federation
transfer variables of the reference classes are control plane and are not reproduced; the PRP seed
is handed in directly.
"""
import os

import numpy as np

from .cipher import FlasheCipher
from .quantize import QuantizingClient

_RNG_RUN_MAX = 1 << 26          # draws per device call of quantize_encrypt (512 MiB of float64)
_DOWNLOAD_WINDOW = 1 << 30      # bytes of unquantised layers decrypt_unquantize keeps on the device before it downloads them

__all__ = ["dynamic_masking_choice", "FlasheClient"]


def dynamic_masking_choice(masks, total):
    """Arbiter.dynamic_masking's decision (jzf_flashe_block.py:92-112): "single" unless double masking
    would need strictly fewer PRF blocks.  single_cost = 2 * sum(len(mask)); double_cost = 2 * single_cost
    minus 2 per position shared by consecutive clients (their masks cancel)."""
    single_cost = 2 * sum(len(m) for m in masks)
    double_cost = 2 * single_cost
    one_hots = []
    for m in masks:
        oh = np.zeros(total, dtype=np.uint8)
        oh[np.asarray(m, dtype=np.int64)] = 1
        one_hots.append(oh)
    canceled = 0
    for i in range(len(masks) - 1):
        canceled += int((one_hots[i] & one_hots[i + 1]).sum())
    double_cost -= canceled * 2
    return "single" if single_cost <= double_cost else "double"


class FlasheClient(object):
    """`jzf_flashe_block._Client` without the transport: holds a FlasheCipher and forwards to it with the
    reference's method names, so `JZFWeights.encrypted(cipher)` / `.decrypted(cipher)`
    (jzf_weights.py:334-338) can be handed this object unchanged."""

    def __init__(self, args, device=0):
        q = args['quantize']
        self.int_bits = q['int_bits']
        self.batch = q.get('batch')
        self.element_bits = q.get('element_bits')
        self.padding = q.get('padding')
        self.secure = q.get('secure')
        self.precompute = args.get('precompute', {}).get('enable', False)
        if self.precompute:
            self.num_params = args['precompute']['num_params']
        self.mask = args.get('mask', 'double')
        self.cipher = None
        self.quantizer = None
        self._device = device

    def create_cipher(self, idx, num_clients, prp_seed):
        """What Guest/Host.create_cipher leave behind (:193-244, :287-326): a keyed cipher that knows its
        client index and, with precompute enabled, the masks of iteration 0."""
        self.cipher = FlasheCipher(self.int_bits, device=self._device)
        self.cipher.idx = idx
        self.cipher.set_num_clients(num_clients)
        self.cipher.generate_prp_seed(prp_seed)
        if self.precompute:
            self.cipher.set_num_params(self.num_params)
            self.cipher.prepare_encrypt()
        # the quantiser the reference creates right after the cipher (:229-238, :311-320); num_clients arrives over the wire there
        self.quantizer = QuantizingClient(self.int_bits, None, None, self.batch, self.element_bits, self.padding, self.secure,
                                          device=self._device)
        self.quantizer.num_clients = num_clients
        return self.cipher

    def dynamic_masking(self, choice, masks):
        """Client side of the arbiter hint (:185-191, :278-285)."""
        if not self.mask == "dynamic":
            return
        self.cipher.masking_scheme = choice
        self.cipher.masks = masks

    def encrypt(self, plaintext, device=None):
        """device (new): True keeps the ciphertext in HBM and returns a DeviceVector, see FlasheCipher.encrypt."""
        return self.cipher.encrypt(plaintext, device=device)

    def decrypt(self, ciphertext, device=None):
        return self.cipher.decrypt(ciphertext, device=device)

    def get_idx_list(self):
        return self.cipher.get_idx_list()

    def set_idx_list(self, idx_list):
        self.cipher.set_idx_list(raw_idx_list=idx_list, mode="decrypt")

    def set_iter_index(self, iter_index):
        self.cipher.set_iter_index(iter_index)
        self.quantizer.set_iter(iter_index)

    # the quantiser forwarders of _Client / Guest / Host (:159-163, :254-264)
    def quantize(self, weights):
        if self.quantizer.layer_size_list is None:
            self.quantizer.set_layer_size_list(weights)
        return self.quantizer.quantize(weights)

    def normalize(self, weights):
        if self.quantizer.layer_size_list is None:
            self.quantizer.set_layer_size_list(weights)
        return self.quantizer.normalize(weights)

    def unquantize(self, weights):
        return self.quantizer.unquantize(weights)

    def unnormalize(self, weights):
        return self.quantizer.unnormalize(weights)

    # ---- the client step with nothing on the host in between (new) ------------------------------------------------------------
    def _fusable(self):
        c = self.cipher
        return (not self.batch and c.masks is None and c.prp_seed is not None and not c.next_iter_encrypt_prepared
                and hasattr(c.engine, "quantize_encrypt_dev"))

    def quantize_encrypt(self, weights, device=True):
        """`self.quantize(weights)` followed by `weights.encrypted(self)` -- QuantizingClient.quantize (jzf_quantize.py:394-491), then
        JZFWeights.encrypted -> _Client.encrypt for every layer in walking order (jzf_weights.py:334-338, :446-450,
        jzf_flashe_block.py:142-150) -- with no host round trip in between: a layer goes up once as it is (4 or 8 bytes per value), its
        stochastic-rounding draws are generated on the device from NumPy's own stream (small layers: drawn on the host), ONE launch
        quantises and encrypts (flashe_quantize_encrypt_dev).  Bit-identical to the two calls, layer by layer, with the same seed.
        Layers come back as DeviceVectors (device=True: the ciphertext stays in HBM for `aggregate`) or as uint64 limb arrays [n, L];
        their shapes are kept for `decrypt_unquantize`.  Batched quantisation, sparse masks and precomputed encrypt masks take the
        two-call path and return what it returns."""
        from . import cipher as _cipher_mod
        from .engine import DeviceVector
        from .quantize import ACIQ, DEVICE_RNG_MIN, _loop_dtype
        q, c = self.quantizer, self.cipher
        if q.layer_size_list is None:
            q.set_layer_size_list(weights)
        if not self._fusable():
            weights = self.quantize(weights)
            for k in weights.walking_order:
                weights._weights[k] = self.cipher.encrypt(weights._weights[k])
            return weights
        eng = c.engine
        aciq = ACIQ(q.element_bits)
        alphas = []
        for i, _size in enumerate(q.layer_size_list):
            a = aciq.get_alpha_gaus_direct(q.past_layer_std_list[i])
            alphas.append(0.1 if a == 0 else a)
        q.r_max_list, q.alpha_list = [], []
        self._layer_shapes = {}
        c.set_idx_list(mode="encrypt")
        scheme = 1 if c.masking_scheme == "double" else 0
        # the stochastic-rounding draws of consecutive layers are ONE stretch of NumPy's stream (np.random.random(layer.shape) per layer in
        # walking order, jzf_quantize.py:55-67 under :417-462): a run of layers is drawn by one device call and every layer takes its
        # slice -- one state round trip per run instead of one per layer.  Runs are capped so the draws of a huge model stay bounded.
        order = list(weights.walking_order)
        sizes = [int(np.asarray(weights._weights[k]).size) for k in order]
        dev_rng = os.environ.get("FLASHE_DEVICE_RNG", "1") != "0" and np.random.get_state()[0] == "MT19937"
        runs, at = {}, 0                                             # first layer of a run -> (layers in the run, draws)
        while at < len(order):
            end, tot = at, 0
            while end < len(order) and (end == at or tot + sizes[end] <= _RNG_RUN_MAX):
                tot += sizes[end]
                end += 1
            runs[at] = (end - at, tot)
            at = end
        layer_cnt = 0
        du_run, u_off = None, 0
        for li, k in enumerate(order):
            if k == 'zzz':
                alpha = 1.0
            else:
                alpha = alphas[layer_cnt]
                q.r_max_list.append(alpha * q.num_clients)
                q.alpha_list.append(alpha)
            layer = np.asarray(weights._weights[k])
            self._layer_shapes[k] = layer.shape
            flat = np.ascontiguousarray(layer).reshape(-1)
            if flat.dtype not in (np.float32, np.float64):
                flat = flat.astype(np.float64)
            want = _loop_dtype(flat.dtype, alpha)
            if flat.dtype != want:
                flat = flat.astype(want)
            n = int(flat.size)
            if li in runs:
                du_run, u_off = None, 0
                if dev_rng and runs[li][1] >= DEVICE_RNG_MIN:
                    du_run = eng.numpy_random_dev(runs[li][1])
            dx = eng.upload(flat)
            if du_run is not None:
                du = du_run.ptr + 8 * u_off
                u_off += n
            else:
                du = eng.upload(np.random.random(layer.shape).reshape(-1))
            ct = DeviceVector(eng, n)
            eng.quantize_encrypt_dev(c.iter_index, c.idx, scheme, n, _cipher_mod.N_JOBS, dx, flat.dtype == np.float64, float(alpha),
                                     q.element_bits, du, ct.buf)
            weights._weights[k] = ct.mark_ready() if device else ct.to_host()
            layer_cnt += 1
        return weights

    def decrypt_unquantize(self, weights):
        """`weights.decrypted(self)` followed by `self.unquantize(weights)` (jzf_weights.py:334-335 -> _Client.decrypt, then
        QuantizingClient.unquantize, jzf_quantize.py:493-540) as ONE launch per layer (flashe_decrypt_unquantize_dev): the aggregate
        -- a DeviceVector, uint64 limbs or object ints -- is decrypted with the prefixes `set_idx_list` left behind and comes back as
        the unquantised float64 layer, reshaped as `quantize_encrypt` saw it.  Precomputed decrypt masks, sparse masks and batched
        values take the two-call path."""
        from . import cipher as _cipher_mod
        from .engine import DeviceVector
        q, c = self.quantizer, self.cipher
        fus = (not self.batch and c.masks is None and c.prp_seed is not None and not c.next_iter_decrypt_prepared
               and hasattr(c.engine, "decrypt_unquantize_dev"))
        if not fus:
            for k in weights.walking_order:
                v = self.cipher.decrypt(weights._weights[k], device=False)
                weights._weights[k] = v
            return self.unquantize(weights)
        eng = c.engine
        if c.masking_scheme == "double":
            add_idx = [c._idx_of(p) for p in (c.index_prefix_for_add or [])]
            minus_idx = [c._idx_of(p) for p in (c.index_prefix_for_minus or [])]
            if not add_idx and not minus_idx:
                raise KeyError('add')
        else:
            add_idx, minus_idx = [], [c._idx_of(p) for p in c.index_prefix_for_minus]
        shapes = getattr(self, "_layer_shapes", {})
        pending, held = [], 0                # launches run ahead of the downloads: a layer comes down while the next ones are computed

        def drain():
            nonlocal held
            for k_, n_, dout_, _dv in pending:
                out = dout_.download(np.float64, n_)
                weights._weights[k_] = out.reshape(shapes.get(k_, out.shape))
            pending.clear()
            held = 0

        for layer_cnt, k in enumerate(weights.walking_order):
            alpha = q.alpha_list[layer_cnt] if k != 'zzz' else 1.0
            v = weights._weights[k]
            if not isinstance(v, DeviceVector):
                v = np.asarray(v)
                if v.dtype == object:
                    v = v.reshape(-1)
            dv, _kind = c._on_device(v, full_width=True)
            n = len(dv)
            dout = eng.alloc(max(8 * n, 16))
            eng.decrypt_unquantize_dev(c.iter_index, add_idx, minus_idx, n, _cipher_mod.N_JOBS, dv.buf, float(alpha), q.element_bits,
                                       q.num_clients, dout)
            pending.append((k, n, dout, dv))
            held += 8 * n
            if held >= _DOWNLOAD_WINDOW:
                drain()
        drain()
        return weights

    def prepare_encrypt(self):
        if self.precompute:
            self.cipher.prepare_encrypt()

    def prepare_decrypt(self):
        if self.precompute:
            self.cipher.prepare_decrypt()
