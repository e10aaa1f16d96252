"""Test double for flashe_amd.engine.Engine backed by the CPU ORACLE.

Lives under tests/ on purpose: it lets the CPU-only suite exercise the *host logic* of
flashe_amd.cipher.FlasheCipher (state machine, prefix selection, precompute caches,
conversions, sparse bookkeeping) against the golden fixtures without a GPU.  The product
package never imports it.
"""
import numpy as np

from oracle import flashe_oracle as orc


class FakeBuf:
    def __init__(self, arr=None, nbytes=0):
        self.arr = arr
        self.nbytes = nbytes if arr is None else arr.nbytes

    def upload(self, arr):
        self.arr = np.ascontiguousarray(arr).copy()
        return self

    def download(self, dtype=np.uint64, count=None):
        flat = np.ascontiguousarray(self.arr).view(dtype).reshape(-1)
        return flat[:count].copy() if count is not None else flat.copy()


class OracleEngine:
    def __init__(self, key, int_bits, device=0, stream=None):
        self.key, self.int_bits, self.limbs = bytes(key), int_bits, 2 if int_bits > 64 else 1

    def set_key(self, key):
        self.key = bytes(key)

    def alloc_vec(self, n, limbs=None):
        return FakeBuf(np.zeros((n, limbs or self.limbs), dtype=np.uint64))

    def alloc(self, nbytes):
        return FakeBuf(np.zeros(max(nbytes // 8, 1), dtype=np.uint64))

    def upload(self, arr):
        return FakeBuf().upload(arr)

    def sync(self):
        pass

    def _sch(self, s):
        return "double" if s == 1 else "single"

    # host-array API
    def mask(self, it, idx_list, n, n_jobs):
        return orc.mask_sum(self.key, it, idx_list, n, n_jobs, self.int_bits)

    def encrypt(self, it, idx, scheme, n_jobs, pt):
        return orc.encrypt(self.key, it, idx, self._sch(scheme), n_jobs, self.int_bits, pt)

    def decrypt(self, it, add_idx, minus_idx, n_jobs, ct):
        return orc.decrypt(self.key, it, add_idx, minus_idx, n_jobs, self.int_bits, ct)

    def aggregate_elem(self, cts):
        return orc.aggregate_elem(cts, self.int_bits)

    # device API on FakeBufs
    def encrypt_dev(self, it, idx, scheme, n, n_jobs, pt, pt_limbs, ct):
        ct.arr = orc.encrypt(self.key, it, idx, self._sch(scheme), n_jobs, self.int_bits, np.ascontiguousarray(pt.arr).reshape(n, pt_limbs))

    def aggregate_elem_dev(self, cts, n, out):
        out.arr = orc.aggregate_elem([np.ascontiguousarray(c.arr).reshape(n, self.limbs) for c in cts], self.int_bits)

    def mask_dev(self, it, idx_list, n, n_jobs, out):
        out.arr = orc.mask_sum(self.key, it, idx_list, n, n_jobs, self.int_bits)

    # ctx-resident mask precompute (flashe_prepare_* / flashe_*_prepared_dev)
    PREPARED_ENCRYPT, PREPARED_DECRYPT = 1, 2

    def prepare_encrypt(self, it_next, idx, scheme, num_params, n_jobs):
        self._prep = getattr(self, "_prep", {})
        self._prep[1] = (orc.mask_sum(self.key, it_next, [idx], num_params, n_jobs, self.int_bits),
                         orc.mask_sum(self.key, it_next, [idx + 1], num_params, n_jobs, self.int_bits) if scheme == 1 else None)

    def prepare_decrypt(self, it, num_clients, num_params, n_jobs):
        self._prep = getattr(self, "_prep", {})
        self._prep[2] = (orc.mask_sum(self.key, it, [num_clients], num_params, n_jobs, self.int_bits), orc.mask_sum(self.key, it, [0], num_params, n_jobs, self.int_bits))

    def prepared_download(self, which, part):
        ent = getattr(self, "_prep", {}).get(which)
        return None if ent is None else ent[0 if part == "add" else 1]

    def prepared_discard(self, which):
        for w in (1, 2):
            if which & w:
                getattr(self, "_prep", {}).pop(w, None)

    def encrypt_prepared_dev(self, n, pt, pt_limbs, ct):
        add, minus = self._prep[1]                              # KeyError = no cache, as the ABI's FLASHE_EINVAL
        assert len(add) == n
        ct.arr = orc.combine(self.int_bits, np.ascontiguousarray(pt.arr).reshape(n, pt_limbs), add, minus)
        del self._prep[1]                                       # consumed

    def decrypt_prepared_dev(self, it, add_idx, minus_idx, n, n_jobs, inp, out):
        add, minus = self._prep[2]
        assert len(add) == n
        v = orc.combine(self.int_bits, np.ascontiguousarray(inp.arr).reshape(n, self.limbs), add, minus)
        if add_idx or minus_idx:
            v = orc.decrypt(self.key, it, add_idx, minus_idx, n_jobs, self.int_bits, v)
        out.arr = v
        del self._prep[2]

    def combine_dev(self, n, inp, in_limbs, add, minus, out):
        a = np.ascontiguousarray(inp.arr).reshape(n, in_limbs)
        out.arr = orc.combine(self.int_bits, a, add.arr if add is not None else None,
                              minus.arr if minus is not None else None)

    def decrypt_dev(self, it, add_idx, minus_idx, n, n_jobs, inp, out):
        out.arr = orc.decrypt(self.key, it, add_idx, minus_idx, n_jobs, self.int_bits, inp.arr.reshape(n, self.limbs))

    def pack_dev(self, n, inp, out):
        out.arr = orc.pack(inp.arr.reshape(n, self.limbs), self.int_bits)

    def unpack_dev(self, n, inp, out):
        out.arr = orc.unpack(inp.arr, n, self.int_bits)

    def aggregate_packed_dev(self, packed, n_limbs, total_bits, out):
        out.arr = orc.aggregate_packed([p.arr[:n_limbs] for p in packed], total_bits)

    def sparse_minus_mask_dev(self, it, locs, ks, total, n_jobs, out, sorted_lists=False):
        out.arr = orc.sparse_minus_mask(self.key, it, [np.asarray(l.arr)[:k] for l, k in zip(locs, ks)], total, n_jobs, self.int_bits)

    def sparse_decrypt_dev(self, it, locs, ks, total, n_jobs, agg, out, sorted_lists=False, bounds=None):
        mask = orc.sparse_minus_mask(self.key, it, [np.asarray(l.arr)[:k] for l, k in zip(locs, ks)], total, n_jobs, self.int_bits)
        out.arr = orc.combine(self.int_bits, np.asarray(agg.arr).reshape(total, -1), None, mask)

    def sparse_double_masks_dev(self, it, locs, ks, total, add_out, minus_out):
        # the reference's own formulation: one-hot vectors, run analysis (jzf_flashe.py:388-407), dense-position masks
        ohs = []
        for l in locs:
            a = np.zeros(total, dtype=np.uint8)
            a[np.asarray(l.arr, dtype=np.int64)] = 1
            ohs.append(a)
        C = len(ohs)
        minus = [ohs[c] & (1 - ohs[c - 1]) if c > 0 else ohs[c] for c in range(C)]
        add = [np.zeros(total, dtype=np.uint8)] + [ohs[c] & (1 - ohs[c + 1]) if c < C - 1 else ohs[c] for c in range(C)]
        add_out.arr = orc.sparse_dense_mask(self.key, it, add, total, self.int_bits)
        minus_out.arr = orc.sparse_dense_mask(self.key, it, minus, total, self.int_bits)

    def sparse_dense_mask_dev(self, it, sels, total, out):
        out.arr = orc.sparse_dense_mask(self.key, it, [s.arr for s in sels], total, self.int_bits)
