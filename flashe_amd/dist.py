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

torch.distributed is plumbing only (process group, all_to_all_single / all_gather over
RCCL, or gloo in the CPU tests); all arithmetic is done by the `ops` object -- `HipOps` over
the C ABI in production.
"""
import torch
import torch.distributed as dist

from .engine import SCHEME_DOUBLE, limbs_of, telescope

__all__ = ["HipOps", "ShardedRound", "slice_len"]

ALIGN = 64          # slice boundaries are multiples of this many elements


def slice_len(n, world):
    per = (n + world - 1) // world
    return ((per + ALIGN - 1) // ALIGN) * ALIGN


class HipOps:
    """torch CUDA tensors (int64 views of limb vectors) -> C-ABI `_dev` calls.  The engine must
    have been created on torch's current stream so that RCCL and the kernels are ordered.
    `side` is an optional second engine (same key / int_bits, its own stream) on which the
    HBM-bound reduce of the pipelined round runs next to the AES-bound kernels."""

    def __init__(self, engine, side=None):
        self.engine = engine
        self.side = side
        self._events = {}

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

    @staticmethod
    def _p(t, elem_offset=0, limbs=1):
        return t.data_ptr() + elem_offset * limbs * 8

    def encrypt(self, it, idx, scheme, n, n_jobs, pt, pt_limbs, ct):
        self.engine.encrypt_dev(it, idx, scheme, n, n_jobs, pt.data_ptr(), pt_limbs, ct.data_ptr())

    def aggregate(self, ptrs_tensors, n, out):
        self.engine.aggregate_elem_dev([t.data_ptr() if torch.is_tensor(t) else t for t in ptrs_tensors], n, out.data_ptr())

    def aggregate_slices(self, buf, n_slices, slice_elems, out):
        L = self.engine.limbs
        self.engine.aggregate_elem_dev([buf.data_ptr() + g * slice_elems * L * 8 for g in range(n_slices)],
                                       slice_elems, out.data_ptr())

    def decrypt_range(self, it, add_idx, minus_idx, n, n_jobs, first, count, inp, out):
        self.engine.decrypt_range_dev(it, add_idx, minus_idx, n, n_jobs, first, count, inp.data_ptr(), out.data_ptr())


class ShardedRound:
    """One FLASHE round (encrypt x clients -> aggregate -> decrypt) with clients sharded over ranks.

    rank r owns clients r*clients_per_rank .. (r+1)*clients_per_rank - 1 (cipher idx = global
    client number), so `world * clients_per_rank` ciphertext vectors are produced per round."""

    def __init__(self, ops, n, int_bits, clients_per_rank, n_jobs, device, rank=0, world=1, group=None,
                 scheme=SCHEME_DOUBLE):
        self.ops, self.n, self.b, self.cpr, self.n_jobs = ops, n, int_bits, clients_per_rank, n_jobs
        self.rank, self.world, self.group, self.scheme = rank, world, group, scheme
        self.L = limbs_of(int_bits)
        self.slice = slice_len(n, world)
        self.padded = self.slice * world
        kw = dict(dtype=torch.int64, device=device)
        self.ct = [torch.zeros(n * self.L, **kw) for _ in range(clients_per_rank)]
        self.partial = torch.zeros(self.padded * self.L, **kw)       # local aggregate, padded to world slices
        self.recv = torch.zeros(self.padded * self.L, **kw) if world > 1 else None
        self.agg_slice = torch.zeros(self.slice * self.L, **kw)
        self.dec_slice = torch.zeros(self.slice * self.L, **kw)
        self.result = torch.zeros(self.padded * self.L, **kw)        # plaintext aggregate on every rank
        self.first = min(rank * self.slice, n)
        self.count = min(self.slice, n - self.first)

    def total_clients(self):
        return self.cpr * self.world

    def encrypt_phase(self, it, pts, pt_limbs):
        """Every local client encrypts its vector (cipher idx = global client number)."""
        for c in range(self.cpr):
            self.ops.encrypt(it, self.rank * self.cpr + c, self.scheme, self.n, self.n_jobs, pts[c], pt_limbs, self.ct[c])

    def aggregate_phase(self):
        """Local C-way mod-add, then (world > 1) the reduce-scatter: all-to-all of slices + local mod-add."""
        self.ops.aggregate(self.ct, self.n, self.partial)
        if self.world > 1:
            dist.all_to_all_single(self.recv, self.partial, group=self.group)
            self.ops.aggregate_slices(self.recv, self.world, self.slice, self.agg_slice)

    def decrypt_phase(self, it):
        """Decrypt the owned slice with the telescoped prefixes of all uploaded clients, then all-gather."""
        uploaded = list(range(self.total_clients()))
        if self.scheme == SCHEME_DOUBLE:
            add_idx, minus_idx = telescope(uploaded)
        else:
            add_idx, minus_idx = [], uploaded
        if self.world > 1:
            if self.count > 0:
                self.ops.decrypt_range(it, add_idx, minus_idx, self.n, self.n_jobs, self.first, self.count,
                                       self.agg_slice, self.dec_slice)
            dist.all_gather_into_tensor(self.result, self.dec_slice, group=self.group)
        else:
            self.ops.decrypt_range(it, add_idx, minus_idx, self.n, self.n_jobs, 0, self.n, self.partial, self.result)
        return self.result

    def run_pipelined(self, it, pts, pt_limbs, chunks=4):
        """Single-GPU round with the arbiter reduce hidden under the AES-bound kernels: the last
        client's encrypt is issued chunk by chunk; as soon as chunk q of every ciphertext exists the
        reduce of chunk q runs on the side stream, and the decrypt of chunk q follows it on the main
        stream.  Same arithmetic, same buffers, only the schedule differs."""
        assert self.world == 1
        ops, n = self.ops, self.n
        step = ((n + chunks - 1) // chunks + 1023) // 1024 * 1024
        bounds = [(f, min(step, n - f)) for f in range(0, n, step)]
        for c in range(self.cpr - 1):
            ops.encrypt(it, c, self.scheme, n, self.n_jobs, pts[c], pt_limbs, self.ct[c])
        last = self.cpr - 1
        for q, (first, count) in enumerate(bounds):
            ops.encrypt_range(it, last, self.scheme, n, self.n_jobs, first, count, pts[last], pt_limbs, self.ct[last])
            ops.signal(f"enc{q}")
            ops.wait(f"enc{q}", on_side=True)
            ops.aggregate_range(self.ct, first, count, self.partial)
            ops.signal(f"agg{q}", on_side=True)
        uploaded = list(range(self.cpr))
        add_idx, minus_idx = telescope(uploaded) if self.scheme == SCHEME_DOUBLE else ([], uploaded)
        for q, (first, count) in enumerate(bounds):
            ops.wait(f"agg{q}")
            ops.decrypt_range_at(it, add_idx, minus_idx, n, self.n_jobs, first, count, self.partial, self.result)
        return self.result

    def run(self, it, pts, pt_limbs):
        """pts: this rank's plaintext tensors (one per local client).  Returns the tensor holding the
        decrypted aggregate (first n*L words valid)."""
        self.encrypt_phase(it, pts, pt_limbs)
        self.aggregate_phase()
        return self.decrypt_phase(it)
