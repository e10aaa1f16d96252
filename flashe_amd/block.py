"""Transport-free mirror of the adapter that drives the cipher in the reference
(federatedml/framework/homo/procedure/jzf_flashe_block.py): the `_Client` forwarders (:120-174) and
the arbiter's `dynamic_masking` cost model (:89-117).  Key exchange, uuid sync and the federation
transfer variables of the reference classes are control plane and are not reproduced; the PRP seed
is handed in directly.
"""
import numpy as np

from .cipher import FlasheCipher
from .quantize import QuantizingClient

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

    def prepare_encrypt(self):
        if self.precompute:
            self.cipher.prepare_encrypt()

    def prepare_decrypt(self):
        if self.precompute:
            self.cipher.prepare_decrypt()
