"""Multi-GPU round: client shards over the GPUs of one node, one process per GPU.

Mapping of the reference's application-level collectives: the
arbiter's *gather* of client models + *reduce* in Python (jzf_aggregator.py:292-308, :404-430)
+ *broadcast* of the aggregate (:502-508) become, inside one node:

  1. every rank encrypts its own clients' vectors and mod-adds them locally (no exchange);
  2. ONE exchange step: a reduce-scatter of the per-rank partial aggregates.  RCCL has no
     128-bit integer type, so it is done as an all-to-all of slices (rank g receives slice g
     of every partial; on xGMI every GPU pair has its own link, so all 7 links run
     concurrently instead of a per-link-bound ring) followed by the local C-way mod-add
     kernel over the received slices;
  3. every rank decrypts the slice it owns (PRF counters are position-indexed:
     flashe_decrypt_range_dev) and an all-gather hands the plaintext aggregate to all ranks.

`run_packed` is the same round with the arbiter's PACKED reduce (one n*b-bit integer per model, carries
crossing element boundaries): limb slices instead of element slices plus one small all-gather of
per-slice carry information.

torch.distributed is plumbing only (process group, all_to_all_single / all_gather over
RCCL, or gloo in the CPU tests); all arithmetic is done by the `ops` object -- `HipOps` over
the C ABI in production.
"""
import contextlib

import torch
import torch.distributed as dist

from .engine import SCHEME_DOUBLE, limbs_of, telescope

__all__ = ["HipOps", "ShardedRound", "slice_len"]

ALIGN = 256         # slice and chunk boundaries are multiples of this many elements (256 consecutive PRF counters share 3 bytes)


def slice_len(n, world):
    per = (n + world - 1) // world
    return ((per + ALIGN - 1) // ALIGN) * ALIGN


class HipOps:
    """torch CUDA tensors (int64 views of limb vectors) -> C-ABI `_dev` calls.  The engine must
    have been created on torch's current stream so that RCCL and the kernels are ordered.
    `side` is an optional second engine (same key / int_bits, its own stream) on which the
    HBM-bound reduce of the pipelined round runs next to the AES-bound kernels."""

    def __init__(self, engine, side=None, side_stream=None):
        self.engine = engine
        self.side = side
        self.side_stream = side_stream      # the torch stream the side engine runs on (collectives follow it)
        self._events = {}

    def on_side(self):
        """Context under which torch.distributed collectives are enqueued on the side stream."""
        if self.side_stream is None:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.side_stream)

    # ---- two-stream plumbing for ShardedRound.run_pipelined (no-ops without a side engine) ----
    def _ev(self, name):
        if name not in self._events:
            self._events[name] = self.engine.event()
        return self._events[name]

    def signal(self, name, on_side=False):
        (self.side if on_side else self.engine).record(self._ev(name))

    def wait(self, name, on_side=False):
        (self.side if on_side else self.engine).wait_event(self._ev(name))

    def encrypt_range(self, it, idx, scheme, n, n_jobs, first, count, pt, pt_limbs, ct):
        L = self.engine.limbs
        self.engine.encrypt_range_dev(it, idx, scheme, n, n_jobs, first, count, pt.data_ptr() + first * pt_limbs * 8,
                                      pt_limbs, ct.data_ptr() + first * L * 8)

    def aggregate_range(self, tensors, first, count, out, on_side=True):
        eng = self.side if (on_side and self.side is not None) else self.engine
        off = first * eng.limbs * 8
        eng.aggregate_elem_dev([t.data_ptr() + off for t in tensors], count, out.data_ptr() + off)

    def decrypt_range_at(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out):
        off = first * self.engine.limbs * 8
        self.engine.decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, inp.data_ptr() + off, out.data_ptr() + off)

    def aggregate_slices_side(self, buf, buf_elem_off, n_slices, slice_elems, out, out_elem_off, extra=None):
        """N-way mod-add of the equally sized pieces received by one chunk's all-to-all (side stream);
        extra = (tensor, elem_offset) is one more operand (the precomputed decrypt mask difference)."""
        eng = self.side if self.side is not None else self.engine
        L = eng.limbs
        base = buf.data_ptr() + buf_elem_off * L * 8
        ptrs = [base + g * slice_elems * L * 8 for g in range(n_slices)]
        if extra is not None:
            ptrs.append(extra[0].data_ptr() + extra[1] * L * 8)
        eng.aggregate_elem_dev(ptrs, slice_elems, out.data_ptr() + out_elem_off * L * 8)

    def decrypt_range_side(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, in_elem_off, out, out_elem_off):
        eng = self.side if self.side is not None else self.engine
        L = eng.limbs
        eng.decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, inp.data_ptr() + in_elem_off * L * 8,
                              out.data_ptr() + out_elem_off * L * 8)

    @staticmethod
    def _p(t, elem_offset=0, limbs=1):
        return t.data_ptr() + elem_offset * limbs * 8

    def encrypt(self, it, idx, scheme, n, n_jobs, pt, pt_limbs, ct):
        self.engine.encrypt_dev(it, idx, scheme, n, n_jobs, pt.data_ptr(), pt_limbs, ct.data_ptr())

    def encrypt_batch(self, it, idx_list, scheme, n, n_jobs, pts, pt_limbs, cts):
        self.engine.encrypt_batch_dev(it, idx_list, scheme, n, n_jobs, [t.data_ptr() for t in pts], pt_limbs,
                                      [t.data_ptr() for t in cts])

    def aggregate(self, ptrs_tensors, n, out):
        self.engine.aggregate_elem_dev([t.data_ptr() if torch.is_tensor(t) else t for t in ptrs_tensors], n, out.data_ptr())

    def aggregate_slices(self, buf, n_slices, slice_elems, out):
        L = self.engine.limbs
        self.engine.aggregate_elem_dev([buf.data_ptr() + g * slice_elems * L * 8 for g in range(n_slices)],
                                       slice_elems, out.data_ptr())

    def decrypt_range(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out):
        self.engine.decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, inp.data_ptr(), out.data_ptr())

    def aggregate_decrypt(self, it, add_idx, minus_idx, n, n_jobs, first, count, srcs, agg_out, out, on_side=False):
        """The reduce fused with the decrypt of its result, on elements [first, first + count) of the vector:
        srcs = [(tensor, elem_offset)] operands addressing element `first`; agg_out / out = (tensor, elem_offset) or None."""
        eng = self.side if (on_side and self.side is not None) else self.engine
        L = eng.limbs
        at = lambda x: None if x is None else x[0].data_ptr() + x[1] * L * 8
        eng.aggregate_decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, [at(x) for x in srcs], at(agg_out), at(out))

    def prf_jobs(self, it, n, n_jobs, jobs):
        """jobs: (add_idx, minus_idx, first, count, in_tensor or None, in_word_offset, in_limbs, out_tensor, out_word_offset);
        one launch for all of them (flashe_prf_jobs_dev)."""
        self.engine.prf_jobs_dev(it, n, n_jobs, [
            (a, m, first, count, None if t_in is None else t_in.data_ptr() + 8 * o_in, in_limbs, t_out.data_ptr() + 8 * o_out)
            for a, m, first, count, t_in, o_in, in_limbs, t_out, o_out in jobs])

    # ---- packed reduce (the arbiter's one-big-integer add, jzf_aggregator.py:406-419) ----
    def pack(self, n, src, dst):
        self.engine.pack_dev(n, src.data_ptr(), dst.data_ptr())

    def unpack(self, n, src, dst):
        self.engine.unpack_dev(n, src.data_ptr(), dst.data_ptr())

    def aggregate_packed(self, tensors, limb_offsets, n_limbs, total_bits, out):
        self.engine.aggregate_packed_dev([t.data_ptr() + 8 * o for t, o in zip(tensors, limb_offsets)], n_limbs, total_bits,
                                         out.data_ptr())

    def packed_probe(self, x, n_limbs, info):
        self.engine.packed_probe_dev(n_limbs, x.data_ptr(), info.data_ptr())

    def packed_add_carry(self, x, n_limbs, total_bits, carry_in):
        self.engine.packed_add_carry_dev(n_limbs, total_bits, carry_in, x.data_ptr())


class ShardedRound:
    """One FLASHE round (encrypt x clients -> aggregate -> decrypt) with clients sharded over ranks.

    rank r owns clients r*clients_per_rank .. (r+1)*clients_per_rank - 1 (cipher idx = global
    client number), so `world * clients_per_rank` ciphertext vectors are produced per round."""

    def __init__(self, ops, n, int_bits, clients_per_rank, n_jobs, device, rank=0, world=1, group=None,
                 scheme=SCHEME_DOUBLE, force_collectives=False):
        self.ops, self.n, self.b, self.cpr, self.n_jobs = ops, n, int_bits, clients_per_rank, n_jobs
        self.rank, self.world, self.group, self.scheme = rank, world, group, scheme
        # world == 1 normally skips the exchange; force_collectives runs it anyway (a 1-rank all-to-all /
        # all-gather), which lets a single-GPU box exercise the exact RCCL code path
        self.exchange = world > 1 or force_collectives
        self.L = limbs_of(int_bits)
        self.slice = slice_len(n, world)
        self.padded = self.slice * world
        kw = dict(dtype=torch.int64, device=device)
        # the local ciphertexts are equally spaced in ONE allocation: the fused reduce + decrypt walks them by stride
        stride = (n * self.L + 1) // 2 * 2                          # every vector 16-byte aligned
        self.ct_all = torch.zeros(clients_per_rank * stride, **kw)
        self.ct = [self.ct_all[c * stride:c * stride + n * self.L] for c in range(clients_per_rank)]
        self.partial = torch.zeros(self.padded * self.L, **kw)       # local aggregate, padded to world slices
        self.recv = torch.zeros(self.padded * self.L, **kw) if self.exchange else None
        self.agg_slice = torch.zeros(self.slice * self.L, **kw)
        self.dec_slice = torch.zeros(self.slice * self.L, **kw)
        self.result = torch.zeros(self.padded * self.L, **kw)        # plaintext aggregate on every rank
        self.first = min(rank * self.slice, n)
        self.count = min(self.slice, n - self.first)

    def total_clients(self):
        return self.cpr * self.world

    def encrypt_phase(self, it, pts, pt_limbs):
        """Every local client encrypts its vector (cipher idx = global client number)."""
        idx = [self.rank * self.cpr + c for c in range(self.cpr)]
        self.ops.encrypt_batch(it, idx, self.scheme, self.n, self.n_jobs, pts, pt_limbs, self.ct)

    def aggregate_phase(self):
        """Local C-way mod-add, then (world > 1) the reduce-scatter: all-to-all of slices + local mod-add."""
        self.ops.aggregate(self.ct, self.n, self.partial)
        if self.exchange:
            dist.all_to_all_single(self.recv, self.partial, group=self.group)
            self.ops.aggregate_slices(self.recv, self.world, self.slice, self.agg_slice)

    def decrypt_phase(self, it):
        """Decrypt the owned slice with the telescoped prefixes of all uploaded clients, then all-gather."""
        uploaded = list(range(self.total_clients()))
        if self.scheme == SCHEME_DOUBLE:
            add_idx, minus_idx = telescope(uploaded)
        else:
            add_idx, minus_idx = [], uploaded
        if self.exchange:
            if self.count > 0:
                self.ops.decrypt_range(it, add_idx, minus_idx, self.n, self.n_jobs, self.first, self.count,
                                       self.agg_slice, self.dec_slice)
            dist.all_gather_into_tensor(self.result, self.dec_slice, group=self.group)
        else:
            self.ops.decrypt_range(it, add_idx, minus_idx, self.n, self.n_jobs, 0, self.n, self.partial, self.result)
        return self.result

    def _pipe_buffers(self, chunks):
        """Block-cyclic ownership for the pipelined schedule: the vector is cut into `chunks` contiguous
        chunks of world * sub elements and rank g owns piece g of EVERY chunk, so each chunk's
        all-to-all / all-gather works on one contiguous block."""
        key = ("pipe", chunks)
        if getattr(self, "_pipe_key", None) == key:
            return
        n, W, L = self.n, self.world, self.L
        sub = ((n + chunks * W - 1) // (chunks * W) + ALIGN - 1) // ALIGN * ALIGN
        self.p_sub, self.p_chunk = sub, sub * W
        padded = self.p_chunk * chunks
        kw = dict(dtype=torch.int64, device=self.partial.device)
        self.p_partial = torch.zeros(padded * L, **kw)
        self.p_result = torch.zeros(padded * L, **kw)
        self.p_recv = torch.zeros(padded * L, **kw) if self.exchange else None
        self.p_agg = torch.zeros(chunks * sub * L, **kw) if self.exchange else None
        self.p_dec = torch.zeros(chunks * sub * L, **kw) if self.exchange else None
        self._pipe_key = key

    def run_pipelined(self, it, pts, pt_limbs, chunks=4, batch_events=None):
        """The round with everything after the last client's encrypt hidden under it, chunk by chunk:
        the last client encrypts chunk q on the main stream; on the side stream chunk q is reduced
        locally, reduce-scattered (all-to-all + mod-add of the received pieces), the owned piece is
        decrypted and all-gathered -- while the main stream already encrypts chunk q + 1.
        Same arithmetic as run(); only the schedule and the (block-cyclic) ownership differ."""
        ops, n, L, W = self.ops, self.n, self.L, self.world
        self._pipe_buffers(chunks)
        sub, chunk = self.p_sub, self.p_chunk
        last = self.cpr - 1
        if last:
            if batch_events:                       # (start, stop) engine events bracketing the batched launch
                ops.engine.record(batch_events[0])
            ops.encrypt_batch(it, [self.rank * self.cpr + c for c in range(last)], self.scheme, n, self.n_jobs,
                              pts[:last], pt_limbs, self.ct[:last])
            if batch_events:
                ops.engine.record(batch_events[1])
        uploaded = list(range(self.total_clients()))
        add_idx, minus_idx = telescope(uploaded) if self.scheme == SCHEME_DOUBLE else ([], uploaded)
        for q in range(chunks):
            first = q * chunk
            cnt = max(0, min(chunk, n - first))
            if cnt:
                ops.encrypt_range(it, self.rank * self.cpr + last, self.scheme, n, self.n_jobs, first, cnt,
                                  pts[last], pt_limbs, self.ct[last])
            ops.signal(f"enc{q}")
            ops.wait(f"enc{q}", on_side=True)
            if cnt:
                ops.aggregate_range(self.ct, first, cnt, self.p_partial)
            if self.exchange:
                blk = slice(first * L, (first + chunk) * L)
                with ops.on_side():
                    dist.all_to_all_single(self.p_recv[blk], self.p_partial[blk], group=self.group)
                ops.aggregate_slices_side(self.p_recv, first, W, sub, self.p_agg, q * sub)
                gfirst = first + self.rank * sub
                gcnt = max(0, min(sub, n - gfirst))
                if gcnt:
                    ops.decrypt_range_side(it, add_idx, minus_idx, n, self.n_jobs, gfirst, gcnt, self.p_agg, q * sub,
                                           self.p_dec, q * sub)
                with ops.on_side():
                    dist.all_gather_into_tensor(self.p_result[blk], self.p_dec[q * sub * L:(q + 1) * sub * L], group=self.group)
            ops.signal(f"done{q}", on_side=True)
        for q in range(chunks):
            ops.wait(f"done{q}")
            if not self.exchange:
                # single rank: the decrypt is as heavy as one encrypt, so it stays on the main stream (right
                # behind the encrypts) and only the HBM-bound reduce is hidden on the side stream
                first = q * chunk
                cnt = max(0, min(chunk, n - first))
                if cnt:
                    ops.decrypt_range_at(it, add_idx, minus_idx, n, self.n_jobs, first, cnt, self.p_partial, self.p_result)
        return self.p_result

    def run_fused(self, it, pts, pt_limbs, chunks=4, launch_events=None):
        """Double mask, nobody dropped.  Per chunk, ONE launch on the main stream does all the mask arithmetic:
        the local clients' encrypts of the chunk plus D = term(iter, C) - term(iter, 0) on the piece of the
        chunk this rank publishes -- the reference's prepare_decrypt (jzf_flashe.py:633-666), kept as the
        fused difference.  The side stream then only runs HBM-bound adds and the exchange: the reduce takes D
        as one more operand, so what it writes IS the plaintext aggregate (decrypt's `value + add - minus`,
        :570-571).  Chunk q's reduce / exchange hides under chunk q + 1's launch.  Same results as run().
        launch_events: optional list of `chunks` entries, each None or a (start, stop) engine event pair that
        brackets that chunk's launch."""
        assert self.scheme == SCHEME_DOUBLE, "run_fused needs the double mask (one add / one minus prefix)"
        ops, n, L, W = self.ops, self.n, self.L, self.world
        self._pipe_buffers(chunks)
        sub, chunk = self.p_sub, self.p_chunk
        if getattr(self, "p_dmask", None) is None or self.p_dmask.numel() != (chunks * sub * L if self.exchange else self.p_partial.numel()):
            self.p_dmask = torch.zeros(chunks * sub * L if self.exchange else self.p_partial.numel(), dtype=torch.int64,
                                       device=self.partial.device)
        C, base = self.total_clients(), self.rank * self.cpr
        for q in range(chunks):
            first = q * chunk
            cnt = max(0, min(chunk, n - first))
            jobs = [(base + c, base + c + 1, first, cnt, pts[c], first * pt_limbs, pt_limbs, self.ct[c], first * L)
                    for c in range(self.cpr)]
            if self.exchange:
                gfirst = min(first + self.rank * sub, n)
                jobs.append((C, 0, gfirst, max(0, min(sub, n - gfirst)), None, 0, 0, self.p_dmask, q * sub * L))
            else:
                jobs.append((C, 0, first, cnt, None, 0, 0, self.p_dmask, first * L))
            ev = launch_events[q] if launch_events else None
            if ev:
                ops.engine.record(ev[0])
            ops.prf_jobs(it, n, self.n_jobs, jobs)
            if ev:
                ops.engine.record(ev[1])
            ops.signal(f"enc{q}")
            ops.wait(f"enc{q}", on_side=True)
            if not self.exchange:
                if cnt:
                    ops.aggregate_range(self.ct + [self.p_dmask], first, cnt, self.p_result)
            else:
                if cnt:
                    ops.aggregate_range(self.ct, first, cnt, self.p_partial)
                blk = slice(first * L, (first + chunk) * L)
                with ops.on_side():
                    dist.all_to_all_single(self.p_recv[blk], self.p_partial[blk], group=self.group)
                ops.aggregate_slices_side(self.p_recv, first, W, sub, self.p_dec, q * sub, extra=(self.p_dmask, q * sub))
                with ops.on_side():
                    dist.all_gather_into_tensor(self.p_result[blk], self.p_dec[q * sub * L:(q + 1) * sub * L], group=self.group)
            ops.signal(f"done{q}", on_side=True)
        for q in range(chunks):
            ops.wait(f"done{q}")
        return self.p_result

    def _packed_buffers(self):
        if getattr(self, "k_sl", None) is not None:
            return
        n, W = self.n, self.world
        self.k_bits = n * self.b
        self.k_nl = nl = (self.k_bits + 63) // 64
        # limb slices of the packed integer: rank g owns limbs [g * sl, (g + 1) * sl) (even count: 16-byte accesses)
        self.k_sl = sl = ((nl + W - 1) // W + 1) // 2 * 2
        kw = dict(dtype=torch.int64, device=self.partial.device)
        self.k_packed = [torch.zeros(nl + (nl & 1), **kw) for _ in range(self.cpr)]
        self.k_partial = torch.zeros(W * sl, **kw)          # this rank's packed sum; limbs beyond nl stay zero
        self.k_full = torch.zeros(W * sl, **kw)             # the packed aggregate on every rank
        self.k_agg = torch.zeros(n * self.L, **kw)
        if self.exchange:
            self.k_recv = torch.zeros(W * sl, **kw)
            self.k_rows = torch.zeros(W * (sl + 2), **kw)   # received slices, each with two zero limbs on top
            self.k_sum = torch.zeros(sl + 2, **kw)
            self.k_info = torch.zeros(3, **kw)
            self.k_infos = torch.zeros(3 * W, **kw)

    def run_packed(self, it, pts, pt_limbs):
        """The round as a dense FLASHE job runs it: every client model travels as ONE n*b-bit integer and the
        arbiter adds those integers mod 2^(n*b) (jzf_aggregator.py:406-419), so carries cross element
        boundaries.  Across ranks (packed variant): local packed sum; all-to-all of limb
        slices; rank g adds its W slices one limb wider than the slice, which leaves the slice's carry-out
        in the extra limb; an all-gather of (low limb, all-ones flag, carry-out) per slice lets every rank
        derive its carry-in, including the case where a carry ripples through a whole slice; carry-in
        applied in place; all-gather of the slices.  Every rank then unpacks and decrypts the aggregate, as
        every client of the reference does."""
        ops, n, W = self.ops, self.n, self.world
        self._packed_buffers()
        nl, sl, bits = self.k_nl, self.k_sl, self.k_bits
        self.encrypt_phase(it, pts, pt_limbs)
        for c in range(self.cpr):
            ops.pack(n, self.ct[c], self.k_packed[c])
        ops.aggregate_packed(self.k_packed, [0] * self.cpr, nl, bits, self.k_partial)
        total = self.k_partial
        if self.exchange:
            dist.all_to_all_single(self.k_recv, self.k_partial, group=self.group)
            self.k_rows.view(W, sl + 2)[:, :sl].copy_(self.k_recv.view(W, sl))
            lo = self.rank * sl
            cnt = max(0, min(sl, nl - lo))
            last = lo + sl >= nl                         # the top slice: its carry-out is dropped (mod 2^(n*b))
            rows, offs = [self.k_rows] * W, [g * (sl + 2) for g in range(W)]
            self.k_info.zero_()
            if cnt and last:
                ops.aggregate_packed(rows, offs, cnt, bits - 64 * lo, self.k_sum)
            elif cnt:
                ops.aggregate_packed(rows, offs, sl + 1, 64 * (sl + 1), self.k_sum)
                ops.packed_probe(self.k_sum, sl + 1, self.k_info)
            dist.all_gather_into_tensor(self.k_infos, self.k_info, group=self.group)
            infos = self.k_infos.cpu().numpy().view("uint64").reshape(W, 3)
            carry = 0
            for g in range(self.rank):                   # slices below this one are full, never the top slice
                low, ones, cout = (int(v) for v in infos[g])
                ripples = ones and low + carry >= 1 << 64
                carry = cout + (1 if ripples else 0)
            if cnt and carry:
                ops.packed_add_carry(self.k_sum, cnt, bits - 64 * lo if last else 64 * sl, carry)
            dist.all_gather_into_tensor(self.k_full, self.k_sum[:sl], group=self.group)
            total = self.k_full
        ops.unpack(n, total, self.k_agg)
        uploaded = list(range(self.total_clients()))
        add_idx, minus_idx = telescope(uploaded) if self.scheme == SCHEME_DOUBLE else ([], uploaded)
        ops.decrypt_range(it, add_idx, minus_idx, n, self.n_jobs, 0, n, self.k_agg, self.result)
        return self.result

    def reduce_decrypt_phase(self, it):
        """aggregate_phase + decrypt_phase with the last reduce fused into the decrypt (one pass over its operands):
        without an exchange the C local ciphertexts, with one the W received pieces of the owned slice."""
        uploaded = list(range(self.total_clients()))
        add_idx, minus_idx = telescope(uploaded) if self.scheme == SCHEME_DOUBLE else ([], uploaded)
        n, W = self.n, self.world
        if not self.exchange:
            self.ops.aggregate_decrypt(it, add_idx, minus_idx, n, self.n_jobs, 0, n, [(t, 0) for t in self.ct],
                                       (self.partial, 0), (self.result, 0))
            return self.result
        self.ops.aggregate(self.ct, n, self.partial)
        dist.all_to_all_single(self.recv, self.partial, group=self.group)
        if self.count > 0:
            self.ops.aggregate_decrypt(it, add_idx, minus_idx, n, self.n_jobs, self.first, self.count,
                                       [(self.recv, g * self.slice) for g in range(W)], (self.agg_slice, 0), (self.dec_slice, 0))
        dist.all_gather_into_tensor(self.result, self.dec_slice, group=self.group)
        return self.result

    def run(self, it, pts, pt_limbs):
        """pts: this rank's plaintext tensors (one per local client).  Returns the tensor holding the
        decrypted aggregate (first n*L words valid).  Two PRF launches per round on one GPU: every local
        encrypt, then reduce + decrypt."""
        self.encrypt_phase(it, pts, pt_limbs)
        return self.reduce_decrypt_phase(it)
