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

__all__ = ["dynamic_masking_choice", "FlasheClient"]


def dynamic_masking_choice(masks, total, engine=None):
    """Arbiter.dynamic_masking's decision (jzf_flashe_block.py:92-112): "single" unless double masking would need strictly fewer PRF
    blocks.  single_cost = 2 * sum(len(mask)); double_cost = 2 * single_cost minus 2 per position shared by consecutive clients
    (their masks cancel).  The reference builds a one-hot vector of `total` entries per client and ANDs neighbours; the positions two
    clients share are the intersection of their location SETS, so nothing of size `total` is needed:
      * masks given as `(device buffer, length)` pairs -- strictly increasing uint32 lists already in HBM (what Sparsifier emits), with
        `engine` the Engine they live on: counted on the device (flashe_dynamic_masking_cost_dev), config-5 size in well under 1 ms;
      * host lists / arrays: sorted-set intersection of neighbours (np.intersect1d), O(k log k)."""
    if engine is not None and masks and all(isinstance(m, tuple) for m in masks):
        single_cost, double_cost = engine.dynamic_masking_cost_dev([m[0] for m in masks], [m[1] for m in masks])
        return "single" if single_cost <= double_cost else "double"
    single_cost = 2 * sum(len(m) for m in masks)
    double_cost = 2 * single_cost
    sets = []
    for m in masks:
        a = np.asarray(m, dtype=np.int64).reshape(-1)
        if a.size and (int(a.max()) >= total or int(a.min()) < -total):
            bad = int(a.max()) if int(a.max()) >= total else int(a.min())
            raise IndexError(f"index {bad} is out of bounds for axis 0 with size {total}")
        sets.append(np.unique(np.where(a < 0, a + total, a)))          # (one_hot[mask] = 1: a set; negative indices wrap as in NumPy)
    canceled = 0
    for i in range(len(sets) - 1):
        canceled += int(np.intersect1d(sets[i], sets[i + 1], assume_unique=True).size)
    double_cost -= canceled * 2
    return "single" if single_cost <= double_cost else "double"


def aggregate_sparse_uploads(engine, uploads, locations, total, device=True):
    """The arbiter's side of a sparse round on the device: what `Arbiter.expand_to_dense` (jzf_aggregator.py:150-165: a dense vector filled
    with the upload's LAST element -- the un-encrypted quantised zero -- and the other elements at the client's locations) followed by the
    reduce over the clients (`:419-430`) computes, in one pass without the C dense intermediates (flashe_sparse_aggregate_dev; strictly
    increasing location lists take the LDS-staged form).
      uploads   per client: a DeviceVector of k + 1 elements as FlasheClient.quantize_encrypt leaves it (it stays where it is), uint64 limbs
                [k + 1, L] or object ints [k + 1]
      locations per client: k sorted positions (host integers, or a device buffer of uint32)
    Returns the dense aggregate of `total` elements: a DeviceVector (device=True) or uint64 limbs [total, L]."""
    from .engine import DeviceBuffer, DeviceVector
    lim = engine.limbs
    vals, zeros, ks, locs, sorted_all = [], [], [], [], True
    for up, loc in zip(uploads, locations):
        if isinstance(up, DeviceVector):
            dv = up.widened(engine) if up.compact else up
            k = len(dv) - 1
            dv.wait_on(engine)
            z = dv.buf.download(np.uint64, (k + 1) * lim)[k * lim:]
            vals.append(dv.buf)
        else:
            a = np.asarray(up)
            if a.dtype == object:
                flat = a.reshape(-1)
                a = np.stack([(flat & (2 ** 64 - 1)).astype(np.uint64)] + ([((flat >> 64) & (2 ** 64 - 1)).astype(np.uint64)] if lim == 2 else []), axis=1)
            a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, lim)
            k = a.shape[0] - 1
            z = a[k]
            vals.append(engine.upload(a[:k]) if k else engine.alloc(16))
        zeros.append([int(v) for v in z])
        ks.append(k)
        if isinstance(loc, DeviceBuffer):
            locs.append(loc)
        else:
            la = np.asarray(loc, dtype=np.int64).reshape(-1)
            if la.size != k:
                raise ValueError(f"{la.size} locations for an upload of {k} values")
            sorted_all = sorted_all and bool(np.all(la[1:] > la[:-1]))
            locs.append(engine.upload(la.astype(np.uint32)) if k else engine.alloc(16))
    out = DeviceVector(engine, int(total))
    if total:
        engine.sparse_aggregate_dev(int(total), locs, ks, vals, zeros, out.buf, sorted_lists=sorted_all)
    return out.mark_ready() if device else out.to_host()


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
        self.shape_dict = None           # layer shapes of the flattened model (what the aggregator-side Client keeps, jzf_aggregator.py:648)
        self.fuse = True                 # False: quantize_encrypt / decrypt_unquantize run the reference's sequence call by call
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

    # ---- flatten / unflatten: Client.flatten_weights / unflatten_weights of the aggregator (jzf_aggregator.py:625-671) ------------------
    def flatten_weights(self, weights):
        """Client.flatten_weights (jzf_aggregator.py:625-650): the layers, in walking order, become ONE vector under the first key; the
        shapes (all but the sparse job's 'zzz' layer) are kept in `self.shape_dict` for unflatten_weights.  A reference job encrypts
        this vector, so PRF counters and the int_bits <= 64 chunking run across the layers."""
        parts, shape_dict, first_k = [np.array([])], {}, None
        for k in list(weights.walking_order):
            if first_k is None:
                first_k = k
            layer = np.asarray(weights._weights[k])
            if k != "zzz":
                shape_dict[k] = layer.shape
            parts.append(layer.flatten())
            del weights._weights[k]
        self.shape_dict = shape_dict
        if first_k is not None:
            weights._weights[first_k] = np.concatenate(parts)
        weights.walking_order = sorted(weights._weights.keys(), key=str)
        return weights

    def unflatten_weights(self, weights):
        """Client.unflatten_weights (jzf_aggregator.py:652-671): cut the one vector back into `self.shape_dict`'s layers."""
        only_key = weights.walking_order[0]
        flat = weights._weights[only_key]
        for k, shape in self.shape_dict.items():
            size = int(np.prod(shape))
            weights._weights[k] = flat[:size].reshape(shape)
            flat = flat[size:]
        weights.walking_order = sorted(weights._weights.keys(), key=str)
        return weights

    # ---- the client step with nothing on the host in between (new) ------------------------------------------------------------
    def _fusable(self, weights=None):
        c = self.cipher
        if not (self.fuse and c.prp_seed is not None and not c.next_iter_encrypt_prepared and hasattr(c.engine, "quantize_encrypt_model_dev")):
            return False
        if weights is None or "zzz" not in weights._weights:
            return c.masks is None
        # the sparse job: compact layers + the sparsifier's trailing one-value layer (the masks a previous round's decrypt left in the
        # cipher do not touch the encrypt)
        order = list(weights.walking_order)
        return not self.batch and len(order) > 1 and order[-1] == "zzz" and np.size(weights._weights["zzz"]) == 1

    def quantize_encrypt(self, weights, device=True):
        """What Client.secure_aggregate does between "begin encoding" and "end encryption" (jzf_aggregator.py:721-743):
        `self.quantize(weights)` -> `flatten_weights` -> [sparse job: strip the trailing quantised zero] -> `weights.encrypted(self)`
        -> [re-append it] -- QuantizingClient.quantize (jzf_quantize.py:394-491), then ONE cipher.encrypt over the flattened model
        (jzf_weights.py:334-338 -> jzf_flashe_block.py:142-150), so that element j of the model is masked with PRF counter j whatever
        layer it sits in -- with no host round trip in between: every layer goes up once as it is (4 or 8 bytes per value) into one flat
        device buffer, the stochastic-rounding draws of the whole model are generated on the device from NumPy's own stream (small
        models: drawn on the host), and ONE launch quantises and encrypts (flashe_quantize_encrypt_model_dev: per-layer alpha from a
        device table).  Bit-identical to the reference's sequence with the same seed (tests/golden/clientstep.json).
        The result has the reference's form: one key (the first of the walking order) holding the flattened ciphertext -- a
        DeviceVector (device=True: it stays in HBM for `aggregate`) or uint64 limbs [n, L]; `self.shape_dict` keeps the shapes for
        `decrypt_unquantize`.  BATCHED jobs ("batch": true, several quantised values per ciphertext element, every layer padded to whole
        elements on its own: jzf_quantize.py:436-451, :162-185) are two launches: quantise + batch of the whole model
        (flashe_quantize_batch_model_dev), then the encrypt of the flattened batched vector.  The SPARSE job (compact layers from
        `Client.sparsify` plus the one-value 'zzz' layer, jzf_aggregator.py:717-743) is the same one launch over the compact layers; the
        'zzz' value is quantised on the host with the draw that follows theirs (alpha 1.0, jzf_quantize.py:433-435) and appended
        UN-encrypted: the result holds n + 1 elements.  Batched sparse jobs and precomputed encrypt masks take the same sequence call by
        call on the host and return object arrays like the reference."""
        from . import cipher as _cipher_mod
        from .engine import DeviceVector
        from .quantize import ACIQ, DEVICE_RNG_MIN, _loop_dtype
        q, c = self.quantizer, self.cipher
        if q.layer_size_list is None:
            q.set_layer_size_list(weights)
        if not self._fusable(weights):
            sparse = "zzz" in weights._weights
            weights = self.flatten_weights(self.quantize(weights))
            k0 = weights.walking_order[0]
            flat = weights._weights[k0]
            if sparse:                                                       # jzf_aggregator.py:735-743
                zero_quantized, flat = flat[-1], flat[:-1]
            ct = self.cipher.encrypt(flat)
            weights._weights[k0] = np.append(ct, [zero_quantized]) if sparse else ct
            return weights
        eng = c.engine
        aciq = ACIQ(q.element_bits)
        alphas = []
        for i, _size in enumerate(q.layer_size_list):
            a = aciq.get_alpha_gaus_direct(q.past_layer_std_list[i])
            alphas.append(0.1 if a == 0 else a)
        q.r_max_list, q.alpha_list = [], []
        c.set_idx_list(mode="encrypt")
        scheme = 1 if c.masking_scheme == "double" else 0
        order = list(weights.walking_order)
        zzz = None
        if "zzz" in weights._weights:                       # (_fusable: it closes the walking order)
            zzz = np.asarray(weights._weights["zzz"])
            order = order[:-1]
        host, starts, offs, shape_dict = [], [], [], {}
        n, nbytes = 0, 0
        for li, k in enumerate(order):
            alpha = alphas[li]
            q.r_max_list.append(alpha * q.num_clients)
            q.alpha_list.append(alpha)
            layer = np.asarray(weights._weights[k])
            shape_dict[k] = layer.shape
            flat = np.ascontiguousarray(layer).reshape(-1)
            if flat.dtype not in (np.float32, np.float64):
                flat = flat.astype(np.float64)
            want = _loop_dtype(flat.dtype, alpha)                         # the dtype NumPy's clip / scale arithmetic runs in
            if flat.dtype != want:
                flat = flat.astype(want)
            host.append(flat)
            starts.append(n)
            offs.append(nbytes)
            n += int(flat.size)
            nbytes += (flat.nbytes + 15) & ~15
        xbuf = eng.alloc(max(nbytes, 16))
        for flat, off in zip(host, offs):
            xbuf.upload_at(off, flat)
        dev_rng = os.environ.get("FLASHE_DEVICE_RNG", "1") != "0" and np.random.get_state()[0] == "MT19937"
        if self.batch:
            # quantise + batch of the whole model in ONE launch (per-layer alpha, per-layer zero padding), then the encrypt of the flattened
            # batched vector: the draws are one stretch of NumPy's stream over all VALUES in walking order
            factor = int(np.ceil(np.log2(q.num_clients)))
            field_bits = q.element_bits + factor
            bs = self.int_bits // field_bits
            q.shape_list = [shape_dict[k] for k in order]
            n_elems = sum((int(h.size) + bs - 1) // bs for h in host)
            du = (eng.numpy_random_dev(n) if dev_rng and n >= DEVICE_RNG_MIN else eng.upload(np.random.random(n))) if n else eng.alloc(16)
            pt = eng.alloc_vec(max(n_elems, 1))
            eng.quantize_batch_model_dev([(int(host[li].size), xbuf.ptr + offs[li], q.alpha_list[li], host[li].dtype == np.float64)
                                          for li in range(len(order))], q.element_bits, field_bits, du, n_elems, pt)
            ct = DeviceVector(eng, n_elems)
            if n_elems:
                eng.encrypt_dev(c.iter_index, c.idx, scheme, n_elems, _cipher_mod.N_JOBS, pt, eng.limbs, ct.buf)
            for k in order:
                del weights._weights[k]
            self.shape_dict = {k: ((int(h.size) + bs - 1) // bs,) for k, h in zip(order, host)}      # the batched layers are 1-D (:448)
            if order:
                weights._weights[order[0]] = ct.mark_ready() if device else ct.to_host()
            weights.walking_order = sorted(weights._weights.keys(), key=str)
            return weights
        table = [(starts[li], xbuf.ptr + offs[li], q.alpha_list[li], host[li].dtype == np.float64) for li in range(len(order))]
        ct = DeviceVector(eng, n + (1 if zzz is not None else 0))
        # the draws of consecutive layers are ONE stretch of NumPy's stream (np.random.random(layer.shape) per layer in walking order,
        # jzf_quantize.py:55-67 under :417-462), i.e. flat element j takes draw j: a run of whole layers is drawn by one device call and
        # quantised + encrypted by one launch over its range.  Runs are capped so the draws of a huge model stay bounded.
        at = 0
        while at < len(order):
            end, tot = at, 0
            while end < len(order) and (end == at or tot + host[end].size <= _RNG_RUN_MAX):
                tot += int(host[end].size)
                end += 1
            if tot:
                du = eng.numpy_random_dev(tot) if dev_rng and tot >= DEVICE_RNG_MIN else eng.upload(np.random.random(tot))
                first = starts[at]
                eng.quantize_encrypt_model_dev(c.iter_index, c.idx, scheme, n, _cipher_mod.N_JOBS, first, tot, table, q.element_bits, du,
                                               ct.ptr + first * eng.limbs * 8)
            at = end
        if zzz is not None:
            # the trailing layer: the next draw of the stream, alpha 1.0, not encrypted (:735-743 strips it before and re-appends it after)
            from .quantize import _as_object, _static_quantize_padding_asymmetric
            flat = zzz.flatten()
            want = _loop_dtype(flat.dtype, 1.0)
            if flat.dtype != want:
                flat = flat.astype(want)
            zq = int(_as_object(_static_quantize_padding_asymmetric(flat, 1.0, q.element_bits, device=q._device, as_object=False)).reshape(-1)[0])
            ct.buf.upload_at(n * eng.limbs * 8, np.array([zq & (2 ** 64 - 1), zq >> 64][:eng.limbs], dtype=np.uint64))
            del weights._weights["zzz"]
        for k in order:
            del weights._weights[k]
        self.shape_dict = shape_dict
        if order:
            weights._weights[order[0]] = ct.mark_ready() if device else ct.to_host()
        weights.walking_order = sorted(weights._weights.keys(), key=str)
        return weights

    def decrypt_unquantize(self, weights):
        """What Client.aggregate does between "begin decryption" and "end decoding" (jzf_aggregator.py:881-899): `weights.decrypted(self)`
        (jzf_weights.py:334-335 -> _Client.decrypt) of the ONE flattened aggregate, `unflatten_weights` by `self.shape_dict`, then
        QuantizingClient.unquantize layer by layer (jzf_quantize.py:493-540) -- as ONE launch (flashe_decrypt_unquantize_model_dev): the
        aggregate -- a DeviceVector, uint64 limbs or object ints -- is decrypted with the prefixes `set_idx_list` left behind and comes
        back as unquantised float64 layers.  The SPARSE job (location lists in `cipher.masks`; it sets `self.shape_dict =
        shape_dict_used_for_sparsification` first, :893-894) decrypts on the device with the sparse minus-mask pass and unquantises the
        dense result in a second launch (flashe_unquantize_model_dev).  Precomputed decrypt masks take the same sequence call by call."""
        from . import cipher as _cipher_mod
        from .engine import DeviceVector
        q, c = self.quantizer, self.cipher
        fus = (self.fuse and c.masks is None and c.prp_seed is not None and not c.next_iter_decrypt_prepared
               and hasattr(c.engine, "decrypt_unquantize_model_dev"))
        k0 = weights.walking_order[0]
        from .cipher import _SparseMinus
        prep = c.next_iter_decrypt_prepared
        sparse_dev = (self.fuse and c.masks is not None and c.masking_scheme == "single" and not self.batch and c.prp_seed is not None
                      and set(prep) == {"minus"} and isinstance(prep["minus"], _SparseMinus) and hasattr(c.engine, "unquantize_model_dev"))
        if sparse_dev:
            eng = c.engine
            dec = self.cipher.decrypt(weights._weights[k0], device=True)                 # the sparse minus-mask pass, result in HBM
            dec = c._as_wide(dec)
            n = len(dec)
            sizes = [int(np.prod(shape)) for shape in self.shape_dict.values()]
            if sum(sizes) > n:
                raise ValueError(f"the aggregate has {n} elements, shape_dict describes {sum(sizes)}")
            table, at = [], 0
            for li, size in enumerate(sizes):
                table.append((at, None, q.alpha_list[li], False))
                at += size
            dout = eng.alloc(max(8 * n, 16))
            if n:
                dec.wait_on(eng)
                eng.unquantize_model_dev(n, 0, n, dec.buf, table, q.element_bits, q.num_clients, dout)
            weights._weights[k0] = dout.download(np.float64, n)
            return self.unflatten_weights(weights)
        if not fus:
            res = self.cipher.decrypt(weights._weights[k0], device=False)
            if isinstance(res, np.ndarray) and res.dtype == np.uint64 and res.ndim == 2:
                from .cipher import _from_limbs
                res = _from_limbs(res, "object")          # (limbs in, limbs out: the layer-by-layer sequence works on the reference's object ints)
            weights._weights[k0] = res
            return self.unquantize(self.unflatten_weights(weights))
        eng = c.engine
        if c.masking_scheme == "double":
            add_idx = [c._idx_of(p) for p in (c.index_prefix_for_add or [])]
            minus_idx = [c._idx_of(p) for p in (c.index_prefix_for_minus or [])]
            if not add_idx and not minus_idx:
                raise KeyError('add')
        else:
            add_idx, minus_idx = [], [c._idx_of(p) for p in c.index_prefix_for_minus]
        v = weights._weights[k0]
        if not isinstance(v, DeviceVector):
            v = np.asarray(v)
            if v.dtype == object:
                v = v.reshape(-1)
        dv, _kind = c._on_device(v, full_width=True)
        dv = c._as_wide(dv)                       # (a compact uint32 aggregate: the fused launch reads one-limb vectors)
        n = len(dv)
        if self.batch:
            # decrypt of the flattened batched vector, then unbatch + `[:size]` + unquantise of every layer in ONE launch
            factor = int(np.ceil(np.log2(q.num_clients)))
            field_bits = q.element_bits + factor
            bs = self.int_bits // field_bits
            names = list(self.shape_dict)
            sizes = [int(np.prod(shape)) for shape in q.shape_list]
            if sum((s_ + bs - 1) // bs for s_ in sizes) != n:
                raise ValueError(f"the aggregate has {n} elements, the batched layers describe {sum((s_ + bs - 1) // bs for s_ in sizes)}")
            dec = eng.alloc_vec(max(n, 1))
            if n:
                eng.decrypt_dev(c.iter_index, add_idx, minus_idx, n, _cipher_mod.N_JOBS, dv.buf, dec)
            n_values = sum(sizes)
            dout = eng.alloc(max(8 * n_values, 16))
            eng.unbatch_unquantize_model_dev([(s_, None, q.alpha_list[li], False) for li, s_ in enumerate(sizes)], q.element_bits, field_bits,
                                             q.num_clients, dec, n, dout)
            out = dout.download(np.float64, n_values)
            del weights._weights[k0]
            at = 0
            for name, shape, s_ in zip(names, q.shape_list, sizes):
                weights._weights[name] = out[at:at + s_].reshape(shape)
                at += s_
            weights.walking_order = sorted(weights._weights.keys(), key=str)
            return weights
        sizes = [int(np.prod(shape)) for shape in self.shape_dict.values()]
        if sum(sizes) > n:
            raise ValueError(f"the aggregate has {n} elements, shape_dict describes {sum(sizes)}")
        table, at = [], 0
        for li, size in enumerate(sizes):
            table.append((at, None, q.alpha_list[li], False))
            at += size
        dout = eng.alloc(max(8 * n, 16))
        if n:
            eng.decrypt_unquantize_model_dev(c.iter_index, add_idx, minus_idx, n, _cipher_mod.N_JOBS, 0, n, dv.buf, table, q.element_bits,
                                             q.num_clients, dout)
        out = dout.download(np.float64, n)
        weights._weights[k0] = out
        return self.unflatten_weights(weights)

    def prepare_encrypt(self):
        if self.precompute:
            self.cipher.prepare_encrypt()

    def prepare_decrypt(self):
        if self.precompute:
            self.cipher.prepare_decrypt()
