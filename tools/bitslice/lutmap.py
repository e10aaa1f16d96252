"""3-input LUT technology mapping of Boolean DAGs (XOR/AND/XNOR/NOT networks) for v_bitop3_b32.

A tiny FPGA-style mapper: K = 3 cut enumeration, area-flow selection, then exact-area
refinement by reference counting.  Leaves flagged `scalar` (round-key planes, which live in
SGPRs) are limited to one per LUT because a gfx950 VALU instruction may read one SGPR operand.
"""
import itertools


class Net:
    def __init__(self):
        self.ops = []          # node -> (op, a, b) ; op in {'in', 'xor', 'and', 'xnor', 'not'}
        self.names = []
        self.scalar = []       # True for SGPR-resident inputs
        self.cache = {}

    def inp(self, name, scalar=False):
        self.ops.append(("in", None, None))
        self.names.append(name)
        self.scalar.append(scalar)
        return len(self.ops) - 1

    def gate(self, op, a, b=None):
        if op in ("xor", "and", "xnor") and a > b:
            a, b = b, a
        key = (op, a, b)
        if key in self.cache:
            return self.cache[key]
        self.ops.append(key)
        self.names.append(None)
        self.scalar.append(False)
        self.cache[key] = len(self.ops) - 1
        return self.cache[key]

    def xor(self, a, b):
        return self.gate("xor", a, b)

    def and_(self, a, b):
        return self.gate("and", a, b)

    def xnor(self, a, b):
        return self.gate("xnor", a, b)

    def not_(self, a):
        return self.gate("not", a)

    def rot16(self, a):
        """Swap the two 16-bit halves of a plane register (a non-LUT op: one v_alignbit_b32).  For the
        mapper it is a boundary: its output is a leaf for cuts, its input must be realised."""
        return self.gate("rot16", a)

    def fanins(self, n):
        op, a, b = self.ops[n]
        if op == "in":
            return ()
        if op in ("not", "rot16"):
            return (a,)
        return (a, b)

    def is_rot(self, n):
        return self.ops[n][0] == "rot16"


def eval_cone(net, node, leaves):
    """8-bit truth table of `node` as a function of up to 3 leaves, in v_bitop3_b32's convention:
    result bit = tt[(src0 << 2) | (src1 << 1) | src2], i.e. leaf 0 <-> 0xF0, leaf 1 <-> 0xCC, leaf 2 <-> 0xAA."""
    pat = [0xF0, 0xCC, 0xAA]
    memo = {l: pat[i] for i, l in enumerate(leaves)}

    def ev(n):
        if n in memo:
            return memo[n]
        op, a, b = net.ops[n]
        if op in ("in", "rot16"):
            raise ValueError("cone reaches an input that is not a leaf")
        if op == "not":
            v = ev(a) ^ 0xFF
        elif op == "xor":
            v = ev(a) ^ ev(b)
        elif op == "and":
            v = ev(a) & ev(b)
        elif op == "xnor":
            v = ev(a) ^ ev(b) ^ 0xFF
        memo[n] = v
        return v

    return ev(node)


def map_luts(net, outputs, max_cuts=16, rounds=4):
    """Returns list of (node, leaves tuple, truth table) in topological order covering `outputs`."""
    n_nodes = len(net.ops)
    fanout = [0] * n_nodes
    for n in range(n_nodes):
        for f in net.fanins(n):
            fanout[f] += 1
    for o in outputs:
        fanout[o] += 1

    cuts = [None] * n_nodes
    af = [0.0] * n_nodes
    best = [None] * n_nodes

    def ok(cut):
        return len(cut) <= 3 and sum(1 for l in cut if net.scalar[l]) <= 1

    def cut_af(cut):
        return 1.0 + sum(af[l] for l in cut)

    for n in range(n_nodes):
        fi = net.fanins(n)
        if not fi:
            cuts[n] = [frozenset([n])]
            af[n] = 0.0
            continue
        if net.is_rot(n):
            cuts[n] = [frozenset([n])]
            best[n] = frozenset(fi)
            af[n] = (1.0 + af[fi[0]]) / max(1, fanout[n])
            continue
        cand = set()
        lists = [cuts[f] for f in fi]
        for combo in itertools.product(*lists):
            c = frozenset().union(*combo)
            if ok(c):
                cand.add(c)
        cand = sorted(cand, key=lambda c: (cut_af(c), len(c)))[:max_cuts]
        if not cand:
            raise RuntimeError("no feasible cut")
        best[n] = cand[0]
        af[n] = cut_af(cand[0]) / max(1, fanout[n])
        cuts[n] = cand + [frozenset([n])]

    # reference-counted exact area refinement
    refs = [0] * n_nodes

    def ref(n):
        """ref node n as a mapped LUT root: returns area added."""
        if not net.fanins(n):
            return 0
        refs[n] += 1
        if refs[n] > 1:
            return 0
        return 1 + sum(ref(l) for l in best[n])

    def deref(n):
        if not net.fanins(n):
            return 0
        refs[n] -= 1
        if refs[n] > 0:
            return 0
        return 1 + sum(deref(l) for l in best[n])

    for o in outputs:
        ref(o)
    for _ in range(rounds):
        changed = False
        for n in range(n_nodes):
            if refs[n] == 0 or not net.fanins(n) or net.is_rot(n):
                continue
            cur = best[n]
            for l in cur:                      # take the current cut out
                deref(l)
            best_c, best_a = None, None
            for c in [cur] + [c for c in cuts[n] if c != cur and c != frozenset([n])]:
                a = sum(ref(l) for l in c)     # area this cut would add ...
                for l in c:
                    deref(l)                   # ... measured, then undone
                if best_a is None or a < best_a:
                    best_c, best_a = c, a
            if best_c != cur:
                changed = True
            best[n] = best_c
            for l in best_c:
                ref(l)
        if not changed:
            break

    used = [n for n in range(n_nodes) if refs[n] > 0 and net.fanins(n)]
    luts = []
    for n in used:
        if net.is_rot(n):
            luts.append((n, tuple(best[n]), "rot16"))
            continue
        leaves = tuple(sorted(best[n], key=lambda l: (net.scalar[l], l)))   # scalar leaf last
        luts.append((n, leaves, eval_cone(net, n, leaves)))
    return luts


def simulate_luts(net, luts, input_values, width=32):
    """Evaluate the LUT netlist on integer bit-vectors; returns dict node -> value."""
    mask = (1 << width) - 1
    val = dict(input_values)
    for n, leaves, tt in luts:
        if tt == "rot16":
            x = val[leaves[0]]
            val[n] = ((x >> 16) | (x << 16)) & mask
            continue
        a = [val[l] for l in leaves] + [0] * (3 - len(leaves))
        out = 0
        for m in range(8):
            if (tt >> m) & 1:
                term = mask
                for i in range(3):
                    term &= a[i] if (m >> (2 - i)) & 1 else (a[i] ^ mask)
                out |= term
        val[n] = out & mask
    return val
