"""Ops double for flashe_amd.dist.ShardedRound backed by the CPU oracle (tests only): same method
surface as flashe_amd.dist.HipOps, operating on CPU torch tensors."""
import numpy as np

from flashe_amd.engine import SCHEME_DOUBLE
from oracle import flashe_oracle as orc

KEY = bytes(range(32))


class OracleOps:
    def __init__(self, b):
        self.b = b
        self.L = 2 if b > 64 else 1

    def _v(self, t, n=None):
        a = t.numpy().view(np.uint64)
        return a if n is None else a[: n * self.L].reshape(n, self.L)

    def encrypt(self, it, idx, scheme, n, n_jobs, pt, pt_limbs, ct):
        self._v(ct, n)[:] = orc.encrypt(KEY, it, idx, "double" if scheme == SCHEME_DOUBLE else "single", n_jobs, self.b,
                                        pt.numpy().view(np.uint64).reshape(n, pt_limbs))

    def encrypt_batch(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts):
        for i, pt, ct in zip(idx_list, pts, cts):
            self.encrypt(it, i, scheme, n, n_jobs, pt, pt_limbs, ct)

    def aggregate(self, tensors, n, out):
        self._v(out, n)[:] = orc.aggregate_elem([self._v(t, n) for t in tensors], self.b)

    def aggregate_slices(self, buf, n_slices, slice_elems, out):
        full = self._v(buf, n_slices * slice_elems)
        parts = [full[g * slice_elems:(g + 1) * slice_elems] for g in range(n_slices)]
        self._v(out, slice_elems)[:] = orc.aggregate_elem(parts, self.b)

    def decrypt_range(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out):
        add = orc.mask_sum(KEY, it, add_idx, n, n_jobs, self.b)[first:first + count]
        minus = orc.mask_sum(KEY, it, minus_idx, n, n_jobs, self.b)[first:first + count]
        self._v(out, count)[:] = orc.combine(self.b, self._v(inp, count), add, minus)

    # two-stream plumbing of the pipelined schedule: a CPU double is sequential, so these are no-ops
    def signal(self, name, on_side=False):
        pass

    def wait(self, name, on_side=False):
        pass

    def encrypt_range(self, it, idx, scheme, n, n_jobs, first, count, pt, pt_limbs, ct):
        full = orc.encrypt(KEY, it, idx, "double" if scheme == SCHEME_DOUBLE else "single", n_jobs, self.b,
                           pt.numpy().view(np.uint64).reshape(n, pt_limbs))
        self._v(ct, n)[first:first + count] = full[first:first + count]

    def aggregate_range(self, tensors, first, count, out, on_side=True):
        self._v(out, first + count)[first:first + count] = orc.aggregate_elem(
            [np.ascontiguousarray(self._v(t, first + count)[first:first + count]) for t in tensors], self.b)

    def decrypt_range_at(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out):
        add = orc.mask_sum(KEY, it, add_idx, n, n_jobs, self.b)[first:first + count]
        minus = orc.mask_sum(KEY, it, minus_idx, n, n_jobs, self.b)[first:first + count]
        self._v(out, first + count)[first:first + count] = orc.combine(
            self.b, np.ascontiguousarray(self._v(inp, first + count)[first:first + count]), add, minus)

    def on_side(self):
        import contextlib
        return contextlib.nullcontext()

    def aggregate_slices_side(self, buf, buf_elem_off, n_slices, slice_elems, out, out_elem_off, extra=None):
        full = self._v(buf, buf_elem_off + n_slices * slice_elems)
        parts = [np.ascontiguousarray(full[buf_elem_off + g * slice_elems: buf_elem_off + (g + 1) * slice_elems]) for g in range(n_slices)]
        if extra is not None:
            parts.append(np.ascontiguousarray(self._v(extra[0], extra[1] + slice_elems)[extra[1]:extra[1] + slice_elems]))
        self._v(out, out_elem_off + slice_elems)[out_elem_off:out_elem_off + slice_elems] = orc.aggregate_elem(parts, self.b)

    def decrypt_range_side(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, in_elem_off, out, out_elem_off):
        add = orc.mask_sum(KEY, it, add_idx, n, n_jobs, self.b)[first:first + count]
        minus = orc.mask_sum(KEY, it, minus_idx, n, n_jobs, self.b)[first:first + count]
        src = np.ascontiguousarray(self._v(inp, in_elem_off + count)[in_elem_off:in_elem_off + count])
        self._v(out, out_elem_off + count)[out_elem_off:out_elem_off + count] = orc.combine(self.b, src, add, minus)

    # ---- packed reduce ----
    def pack(self, n, src, dst):
        p = orc.pack(self._v(src, n), self.b)
        dst.numpy().view(np.uint64)[: len(p)] = p

    def unpack(self, n, src, dst):
        nl = (n * self.b + 63) // 64
        self._v(dst, n)[:] = orc.unpack(src.numpy().view(np.uint64)[:nl], n, self.b)

    def aggregate_packed(self, tensors, limb_offsets, n_limbs, total_bits, out):
        parts = [np.ascontiguousarray(t.numpy().view(np.uint64)[o:o + n_limbs]) for t, o in zip(tensors, limb_offsets)]
        out.numpy().view(np.uint64)[:n_limbs] = orc.aggregate_packed(parts, total_bits)

    def packed_probe(self, x, n_limbs, info):
        a = x.numpy().view(np.uint64)
        i = info.numpy().view(np.uint64)
        i[0], i[1], i[2] = a[0], int(bool((a[1:n_limbs - 1] == np.uint64(2 ** 64 - 1)).all())), a[n_limbs - 1]

    def packed_add_carry(self, x, n_limbs, total_bits, carry_in):
        a = x.numpy().view(np.uint64)
        v = (int.from_bytes(a[:n_limbs].tobytes(), "little") + carry_in) % (1 << total_bits)
        a[:n_limbs] = np.frombuffer(v.to_bytes(8 * n_limbs, "little"), dtype=np.uint64)

    def prf_jobs(self, it, n, n_jobs, jobs):
        for a, m, first, count, t_in, o_in, in_limbs, t_out, o_out in jobs:
            if count == 0:
                continue
            add = orc.mask(KEY, it, a, n, n_jobs, self.b)[first:first + count]
            minus = orc.mask(KEY, it, m, n, n_jobs, self.b)[first:first + count] if m is not None else None
            if t_in is None:
                inp = np.zeros((count, self.L), dtype=np.uint64)
            else:
                inp = t_in.numpy().view(np.uint64)[o_in:o_in + count * in_limbs].reshape(count, in_limbs)
            t_out.numpy().view(np.uint64)[o_out:o_out + count * self.L] = orc.combine(self.b, inp, add, minus).reshape(-1)

    def aggregate_decrypt(self, it, add_idx, minus_idx, n, n_jobs, first, count, srcs, agg_out, out, on_side=False):
        parts = [np.ascontiguousarray(self._v(t, o + count)[o:o + count]) for t, o in srcs]
        agg = orc.aggregate_elem(parts, self.b)
        if agg_out is not None:
            self._v(agg_out[0], agg_out[1] + count)[agg_out[1]:agg_out[1] + count] = agg
        add = orc.mask_sum(KEY, it, add_idx, n, n_jobs, self.b)[first:first + count]
        minus = orc.mask_sum(KEY, it, minus_idx, n, n_jobs, self.b)[first:first + count]
        self._v(out[0], out[1] + count)[out[1]:out[1] + count] = orc.combine(self.b, agg, add, minus)
