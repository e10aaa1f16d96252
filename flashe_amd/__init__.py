"""flashe_amd -- MI355X-native FLASHE cipher engine.

One hot path of SamuelGong/FLASHE (federatedml/secureprotol FLASHE cipher + the arbiter's
ciphertext reduce) as hand-written HIP kernels for gfx950 behind a C ABI
(include/flashe.h -> flashe_amd/libflashe_hip.so), with a host-side mirror of the
reference's `FlasheCipher` API.  See DESIGN.md / INTEGRATION.md.
"""
from .cipher import FlasheCipher, aggregate          # noqa: F401
from .engine import Engine, DeviceBuffer, DeviceVector, FlasheError  # noqa: F401
from .block import FlasheClient, aggregate_sparse_uploads, dynamic_masking_choice  # noqa: F401

__all__ = ["FlasheCipher", "aggregate", "Engine", "DeviceBuffer", "DeviceVector", "FlasheError", "FlasheClient", "aggregate_sparse_uploads", "dynamic_masking_choice"]
