"""Ops double for flashe_amd.dist.ShardedRound backed by the CPU oracle (tests only): the method surface of
flashe_amd.dist.HipOps on host buffers (numpy uint64 arrays); the exchange goes over torch.distributed / gloo when a
process group exists (tests/dist_worker.py), and is the identity for one rank."""
import numpy as np

from flashe_amd.engine import SCHEME_DOUBLE
from oracle import flashe_oracle as orc

KEY = bytes(range(32))


class HostBuf:
    def __init__(self, words):
        self.a = np.zeros(int(words), dtype=np.uint64)


class GlooComm:
    """The collectives of flashe_amd.dist.RcclComm over gloo, on numpy buffers."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_to_all(self, send, send_stride, recv, recv_stride, words):
        import torch
        W = self.world
        s = torch.from_numpy(np.stack([send[p * send_stride:p * send_stride + words] for p in range(W)]).view(np.int64).copy())
        r = torch.empty_like(s)
        self.dist.all_to_all_single(r, s)
        got = r.numpy().view(np.uint64)
        for p in range(W):
            recv[p * recv_stride:p * recv_stride + words] = got[p]

    def all_gather(self, send, recv, words):
        import torch
        s = torch.from_numpy(send[:words].view(np.int64).copy())
        r = torch.empty(self.world * words, dtype=torch.int64)
        self.dist.all_gather_into_tensor(r, s)
        recv[: self.world * words] = r.numpy().view(np.uint64)

    def allreduce_modadd(self, buf, words, b):
        import torch
        t = torch.from_numpy(buf[:words].view(np.int64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)                       # int64 wraps mod 2^64
        got = t.numpy().view(np.uint64)
        buf[:words] = got & np.uint64((1 << b) - 1) if b < 64 else got

    def allreduce(self, value, op):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op={0: self.dist.ReduceOp.MAX, 1: self.dist.ReduceOp.MIN, 2: self.dist.ReduceOp.SUM}[op])
        return float(t.item())


class OracleOps:
    def __init__(self, b, comm=None):
        self.b = b
        self.L = 2 if b > 64 else 1
        self.comm = comm
        self.rank = comm.rank if comm else 0
        self.world = comm.world if comm else 1
        self.side = None

    # ---- memory ----
    def alloc(self, words):
        return HostBuf(max(int(words), 2))

    def upload(self, arr):
        raw = np.ascontiguousarray(arr).reshape(-1).view(np.uint8)              # any element type (uint32 location lists too)
        raw = np.concatenate([raw, np.zeros(-raw.size % 8, dtype=np.uint8)]).view(np.uint64)
        buf = HostBuf(max(raw.size, 2))
        buf.a[:raw.size] = raw
        return buf

    def _w(self, ref, words):
        buf, off = ref
        return buf.a[off:off + words]

    def _v(self, ref, count, limbs=None):
        limbs = limbs or self.L
        return self._w(ref, count * limbs).reshape(count, limbs)

    def read(self, ref, words):
        return self._w(ref, words).copy()

    def zero(self, ref, words):
        self._w(ref, words)[:] = 0

    def sync(self):
        pass

    def signal(self, name, side=False):
        pass

    def wait(self, name, side=False):
        pass

    # ---- cipher ----
    def _name(self, scheme):
        return "double" if scheme == SCHEME_DOUBLE else "single"

    def encrypt_batch(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts, sum_out=None):
        for i, pt, ct in zip(idx_list, pts, cts):
            self._v(ct, n)[:] = orc.encrypt(KEY, it, i, self._name(scheme), n_jobs, self.b, np.ascontiguousarray(self._v(pt, n, pt_limbs)))
        if sum_out is not None:
            self._v(sum_out, n)[:] = orc.aggregate_elem([np.ascontiguousarray(self._v(ct, n)) for ct in cts], self.b)

    def encrypt_range(self, it, idx, scheme, n, n_jobs, first, count, pt, pt_limbs, ct):
        # pt / ct address element `first`: rebuild the whole-vector view the oracle wants
        full_pt = np.zeros((n, pt_limbs), dtype=np.uint64)
        full_pt[first:first + count] = self._v(pt, count, pt_limbs)
        full = orc.encrypt(KEY, it, idx, self._name(scheme), n_jobs, self.b, full_pt)
        self._v(ct, count)[:] = full[first:first + count]

    def encrypt_batch_range(self, it, idx_list, scheme, n, n_jobs, first, count, pts, pt_limbs, cts, sum_out=None):
        assert first <= n and count <= n - first
        for i, pt, ct in zip(idx_list, pts, cts):
            self.encrypt_range(it, i, scheme, n, n_jobs, first, count, pt, pt_limbs, ct)
        if sum_out is not None and count:
            self._v(sum_out, count)[:] = orc.aggregate_elem([np.ascontiguousarray(self._v(ct, count)) for ct in cts], self.b)

    def view(self, buf, off_words):
        if not off_words:
            return buf
        v = HostBuf(2)
        v.a = buf.a[off_words:]
        return v

    def _masks(self, it, add_idx, minus_idx, n, n_jobs, first, count):
        return (orc.mask_sum(KEY, it, add_idx, n, n_jobs, self.b)[first:first + count],
                orc.mask_sum(KEY, it, minus_idx, n, n_jobs, self.b)[first:first + count])

    def decrypt_range(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out, side=False):
        add, minus = self._masks(it, add_idx, minus_idx, n, n_jobs, first, count)
        self._v(out, count)[:] = orc.combine(self.b, np.ascontiguousarray(self._v(inp, count)), add, minus)

    def aggregate(self, srcs, count, out, side=False):
        self._v(out, count)[:] = orc.aggregate_elem([np.ascontiguousarray(self._v(r, count)) for r in srcs], self.b)

    def aggregate_decrypt(self, it, add_idx, minus_idx, n, n_jobs, first, count, srcs, agg_out, out, side=False):
        agg = orc.aggregate_elem([np.ascontiguousarray(self._v(r, count)) for r in srcs], self.b)
        if agg_out is not None:
            self._v(agg_out, count)[:] = agg
        add, minus = self._masks(it, add_idx, minus_idx, n, n_jobs, first, count)
        self._v(out, count)[:] = orc.combine(self.b, agg, add, minus)

    def prf_jobs(self, it, n, n_jobs, jobs):
        for a, m, first, count, inp, in_limbs, out in jobs:
            assert first <= n and count <= n - first, "the C ABI rejects a range beyond the vector, empty or not"
            if count == 0:
                continue
            add = orc.mask(KEY, it, a, n, n_jobs, self.b)[first:first + count]
            minus = orc.mask(KEY, it, m, n, n_jobs, self.b)[first:first + count] if m is not None else None
            src = np.zeros((count, self.L), dtype=np.uint64) if inp is None else np.ascontiguousarray(self._v(inp, count, in_limbs))
            self._v(out, count)[:] = orc.combine(self.b, src, add, minus)

    # ---- packed reduce ----
    def pack(self, n, src, dst):
        p = orc.pack(np.ascontiguousarray(self._v(src, n)), self.b)
        self._w(dst, len(p))[:] = p

    def unpack(self, n, src, dst):
        nl = (n * self.b + 63) // 64
        self._v(dst, n)[:] = orc.unpack(np.ascontiguousarray(self._w(src, nl)), n, self.b)

    def aggregate_packed(self, srcs, n_limbs, total_bits, out):
        self._w(out, n_limbs)[:] = orc.aggregate_packed([np.ascontiguousarray(self._w(r, n_limbs)) for r in srcs], total_bits)

    def packed_probe(self, x, n_limbs, info):
        a, i = self._w(x, n_limbs), self._w(info, 3)
        i[0], i[1], i[2] = a[0], int(bool((a[1:n_limbs - 1] == np.uint64(2 ** 64 - 1)).all())), a[n_limbs - 1]

    def packed_resolve_carry(self, x, n_limbs, total_bits, infos, n_below, stride_words=3):
        buf, off = infos
        carry = 0
        for g in range(n_below):
            low, ones, cout = (int(v) for v in buf.a[off + g * stride_words:off + g * stride_words + 3])
            carry = cout + (1 if (ones and low + carry >= 1 << 64) else 0)
        a = self._w(x, n_limbs)
        v = (int.from_bytes(a.tobytes(), "little") + carry) % (1 << total_bits)
        a[:] = np.frombuffer(v.to_bytes(8 * n_limbs, "little"), dtype=np.uint64)

    # ---- exchange ----
    # ---- the sparse round by position ranges (host restatement: whole vectors through the oracle, then the owned range) ----
    def sparse_span(self):
        return 1752                                   # any span size works for the arithmetic; the device's is used so that the ranges match

    def sparse_bounds(self, total, locs, ks, handle=None):
        return "bounds"

    def _list(self, ref, k):
        buf, off = ref
        return buf.a[off:].view(np.uint32)[:k].copy()

    def sparse_encrypt_aggregate(self, it, idx, locs, ks, pts, pt_limbs, zeros, total, n_jobs, cts, agg, bounds, first, count):
        acc = np.zeros((total, self.L), dtype=np.uint64)
        for c, (i, k) in enumerate(zip(idx, ks)):
            loc = self._list(locs[c], k)
            full = orc.encrypt(KEY, it, i, "single", n_jobs, self.b, np.ascontiguousarray(self._v(pts[c], k, pt_limbs))) if k else np.zeros((0, self.L), dtype=np.uint64)
            own = (loc >= first) & (loc < first + count)
            self._v(cts[c], k)[own] = full[own]       # (entries of other ranges: not this rank's to write)
            z = np.array([[int(zeros[c])] + [0] * (self.L - 1)], dtype=np.uint64)
            acc = orc.aggregate_elem([acc, orc.expand_to_dense(total, loc, full, z, self.b)], self.b)
        self._v(agg, count)[:] = acc[first:first + count]

    def sparse_decrypt(self, it, locs, ks, total, n_jobs, agg, out, bounds, first, count):
        mask = orc.sparse_minus_mask(KEY, it, [self._list(l, k) for l, k in zip(locs, ks)], total, n_jobs, self.b)
        self._v(out, count)[:] = orc.combine(self.b, np.ascontiguousarray(self._v(agg, count)), None, np.ascontiguousarray(mask[first:first + count]))

    def all_to_all(self, send, send_stride, recv, recv_stride, words, side=False):
        sb, so = send
        rb, ro = recv
        if self.comm is None:
            rb.a[ro:ro + words] = sb.a[so:so + words]
            return
        self.comm.all_to_all(sb.a[so:], send_stride, rb.a[ro:], recv_stride, words)

    def all_gather(self, send, recv, words, side=False):
        sb, so = send
        rb, ro = recv
        if self.comm is None:
            rb.a[ro:ro + words] = sb.a[so:so + words]
            return
        self.comm.all_gather(sb.a[so:], rb.a[ro:], words)

    def allreduce_modadd(self, ref, words, side=False):
        rb, ro = ref
        if self.comm is None:
            if self.b < 64:
                rb.a[ro:ro + words] &= np.uint64((1 << self.b) - 1)
            return
        self.comm.allreduce_modadd(rb.a[ro:], words, self.b)

    def allreduce(self, value, op=0):
        return float(value) if self.comm is None else self.comm.allreduce(value, op)

    def barrier(self):
        if self.comm is not None:
            self.comm.allreduce(0.0, 2)
