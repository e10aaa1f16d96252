#!/usr/bin/env python3
"""Generates flashe_amd/csrc/bitslice/aes_bitslice_gen.h (or the path given as the first argument; not tracked, `make bitslice`
and tests/test_bitslice_host.py run this generator): bit-sliced AES-256 round functions as straight-line
v_bitop3_b32 (3-input LUT) code, mapped from the verified Boyar-Peralta S-box circuit plus
ShiftRows / MixColumns / AddRoundKey expressed on bit planes.

Plane numbering: plane[8*B + k] holds bit k (0 = LSB) of state byte B (FIPS-197 order: B = 4*col + row)
for 32 blocks (bit p of the 32-bit word = block p).  Round keys are planes too (0 or ~0), read
through a scalar pointer.

The generated code is verified here, before it is written, by simulating the LUT netlist on random
blocks against a byte-wise AES round."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lutmap import Net, map_luts, simulate_luts  # noqa: E402
from sbox_circuit import parse as parse_sbox, sbox_table  # noqa: E402

SBOX_GATES = parse_sbox()
SB = sbox_table()


def add_sbox(net, xbits):
    """xbits[k] = node of bit k (0 = LSB) -> list of 8 output nodes (bit k)."""
    v = {f"x{i}": xbits[7 - i] for i in range(8)}
    for out, op, a, b in SBOX_GATES:
        if op == "^":
            v[out] = net.xor(v[a], v[b])
        elif op == "&":
            v[out] = net.and_(v[a], v[b])
        else:
            v[out] = net.xnor(v[a], v[b])
    return [v[f"s{7 - k}"] for k in range(8)]


def build_round(final):
    """State planes in -> state planes out for one round: SubBytes, ShiftRows, [MixColumns], AddRoundKey."""
    net = Net()
    S = [[net.inp(f"s[{8 * B + k}]") for k in range(8)] for B in range(16)]
    K = [[net.inp(f"rk[{8 * B + k}]", scalar=True) for k in range(8)] for B in range(16)]
    sb = [add_sbox(net, S[B]) for B in range(16)]
    # ShiftRows: out byte (row r, col c) = in byte (row r, col (c + r) % 4)
    sr = [[None] * 8 for _ in range(16)]
    for c in range(4):
        for r in range(4):
            sr[4 * c + r] = sb[4 * ((c + r) % 4) + r]
    outs = [[None] * 8 for _ in range(16)]
    for c in range(4):
        a = [sr[4 * c + r] for r in range(4)]
        if final:
            for r in range(4):
                for k in range(8):
                    outs[4 * c + r][k] = net.xor(a[r][k], K[4 * c + r][k])
            continue
        u = [[net.xor(a[r][k], a[(r + 1) % 4][k]) for k in range(8)] for r in range(4)]
        T = [net.xor(u[0][k], u[2][k]) for k in range(8)]
        for r in range(4):
            for k in range(8):
                # xtime(u)[k] = u[k-1] (k >= 1), plus u[7] for k in {0, 1, 3, 4}
                x = net.xor(a[r][k], T[k])
                if k >= 1:
                    x = net.xor(x, u[r][k - 1])
                if k in (0, 1, 3, 4):
                    x = net.xor(x, u[r][7])
                outs[4 * c + r][k] = net.xor(x, K[4 * c + r][k])
    flat_out = [outs[B][k] for B in range(16) for k in range(8)]
    return net, S, K, flat_out


def build_round_packed(final):
    """Packed variant: plane register 8*Bp + k (Bp = 0..7) holds bit k of state byte Bp in its LOW 16 bits
    and of state byte Bp + 8 in its HIGH 16 bits (16 blocks per lane).  Bytes Bp and Bp + 8 sit in the same
    row of columns c and c + 2, so SubBytes / MixColumns / AddRoundKey act on both halves at once and
    ShiftRows becomes a re-wiring plus half-swaps (rot16)."""
    net = Net()
    S = [[net.inp(f"s[{8 * B + k}]") for k in range(8)] for B in range(8)]
    K = [[net.inp(f"rk[{8 * B + k}]", scalar=True) for k in range(8)] for B in range(8)]
    sb = [add_sbox(net, S[B]) for B in range(8)]
    sr = [[None] * 8 for _ in range(8)]
    for c in range(2):
        for r in range(4):
            t = (c + r) % 4                      # old column feeding the low half of new pair c, row r
            src = sb[4 * (t % 2) + r]
            sr[4 * c + r] = [net.rot16(x) for x in src] if t >= 2 else src
    outs = [[None] * 8 for _ in range(8)]
    for c in range(2):
        a = [sr[4 * c + r] for r in range(4)]
        if final:
            for r in range(4):
                for k in range(8):
                    outs[4 * c + r][k] = net.xor(a[r][k], K[4 * c + r][k])
            continue
        u = [[net.xor(a[r][k], a[(r + 1) % 4][k]) for k in range(8)] for r in range(4)]
        T = [net.xor(u[0][k], u[2][k]) for k in range(8)]
        for r in range(4):
            for k in range(8):
                x = net.xor(a[r][k], T[k])
                if k >= 1:
                    x = net.xor(x, u[r][k - 1])
                if k in (0, 1, 3, 4):
                    x = net.xor(x, u[r][7])
                outs[4 * c + r][k] = net.xor(x, K[4 * c + r][k])
    flat_out = [outs[B][k] for B in range(8) for k in range(8)]
    return net, S, K, flat_out


def verify_packed(net, S, K, flat_out, luts, final, trials=3):
    rnd = random.Random(4321 + final)
    for _ in range(trials):
        blocks = [[rnd.randrange(256) for _ in range(16)] for _ in range(16)]
        rk = [rnd.randrange(256) for _ in range(16)]
        vals = {}
        for B in range(8):
            for k in range(8):
                lo = sum(((blocks[p][B] >> k) & 1) << p for p in range(16))
                hi = sum(((blocks[p][B + 8] >> k) & 1) << p for p in range(16))
                vals[S[B][k]] = lo | (hi << 16)
                vals[K[B][k]] = (0xFFFF if (rk[B] >> k) & 1 else 0) | (0xFFFF0000 if (rk[B + 8] >> k) & 1 else 0)
        res = simulate_luts(net, luts, vals)
        for p in range(16):
            want = ref_round(blocks[p], rk, final)
            got = [0] * 16
            for B in range(8):
                for k in range(8):
                    v = res[flat_out[8 * B + k]]
                    got[B] |= ((v >> p) & 1) << k
                    got[B + 8] |= ((v >> (16 + p)) & 1) << k
            assert got == want, (final, p)


def emit_packed(net, luts, flat_out, fname, S, K):
    by_node = {n: (n, leaves, tt) for n, leaves, tt in luts}
    done, order, marks = set(), [], []

    def visit(n):
        if n in done or n not in by_node:
            return
        for l in by_node[n][1]:
            visit(l)
        done.add(n)
        order.append(by_node[n])

    for c in range(2):
        for r in range(4):
            for k in range(8):
                visit(flat_out[8 * (4 * c + r) + k])
        marks.append(len(order))
    assert len(order) == len(luts)
    name = {}
    for B in range(8):
        for k in range(8):
            name[S[B][k]] = f"s[{8 * B + k}]"
            name[K[B][k]] = f"rk[{8 * B + k}]"
    lines = []
    n_lut = n_rot = 0
    for pos, (n, leaves, tt) in enumerate(order):
        if pos in marks:
            lines.append("    __builtin_amdgcn_sched_barrier(0);")
        name[n] = f"t{n}"
        args = [name[l] for l in leaves]
        if tt == "rot16":
            expr = f"rot16({args[0]})"
            n_rot += 1
        elif len(leaves) == 3:
            expr = f"lut3<0x{tt:02x}>({args[0]}, {args[1]}, {args[2]})"
            n_lut += 1
        elif len(leaves) == 2:
            simple = {0x3c: f"({args[0]} ^ {args[1]})", 0xc0: f"({args[0]} & {args[1]})",
                      0xc3: f"~({args[0]} ^ {args[1]})", 0xfc: f"({args[0]} | {args[1]})"}
            expr = simple.get(tt, f"lut3<0x{tt:02x}>({args[0]}, {args[1]}, {args[1]})")
            n_lut += 1
        else:
            assert tt in (0xf0, 0x0f), hex(tt)
            expr = f"~{args[0]}" if tt == 0x0f else args[0]
            n_lut += 1
        lines.append(f"    const uint32_t {name[n]} = {expr};")
    for i, n in enumerate(flat_out):
        lines.append(f"    o[{i}] = {name[n]};")
    body = "\n".join(lines)
    return (f"// {n_lut} LUT ops + {n_rot} half-swaps\n"
            f"__device__ __forceinline__ void {fname}(const uint32_t (&s)[64], const uint32_t *__restrict__ rk, uint32_t (&o)[64])\n"
            "{\n" + body + "\n}\n"), n_lut, n_rot


# ---------------------------------------------------------------- reference round on bytes
def xtime(a):
    a <<= 1
    return (a ^ 0x1b) & 0xff if a & 0x100 else a


def ref_round(state, rk, final):
    s = [SB[b] for b in state]
    t = [0] * 16
    for c in range(4):
        for r in range(4):
            t[4 * c + r] = s[4 * ((c + r) % 4) + r]
    if not final:
        o = [0] * 16
        for c in range(4):
            a = t[4 * c:4 * c + 4]
            for r in range(4):
                o[4 * c + r] = xtime(a[r]) ^ (xtime(a[(r + 1) % 4]) ^ a[(r + 1) % 4]) ^ a[(r + 2) % 4] ^ a[(r + 3) % 4]
        t = o
    return [t[i] ^ rk[i] for i in range(16)]


def verify(net, S, K, flat_out, luts, final, trials=3):
    rnd = random.Random(1234 + final)
    for _ in range(trials):
        blocks = [[rnd.randrange(256) for _ in range(16)] for _ in range(32)]
        rk = [rnd.randrange(256) for _ in range(16)]
        vals = {}
        for B in range(16):
            for k in range(8):
                vals[S[B][k]] = sum(((blocks[p][B] >> k) & 1) << p for p in range(32))
                vals[K[B][k]] = 0xFFFFFFFF if (rk[B] >> k) & 1 else 0
        res = simulate_luts(net, luts, vals)
        for p in range(32):
            want = ref_round(blocks[p], rk, final)
            got = [sum(((res[flat_out[8 * B + k]] >> p) & 1) << k for k in range(8)) for B in range(16)]
            assert got == want, (final, p)


def cone_order(net, luts, flat_out):
    """Reorder LUTs so that each output column's cone (its 4 S-boxes, then its MixColumns) is emitted
    contiguously: the inputs of a column die as its outputs appear, keeping ~128 planes live."""
    by_node = {n: (n, leaves, tt) for n, leaves, tt in luts}
    done, order, marks = set(), [], []

    def visit(n):
        if n in done or n not in by_node:
            return
        for l in by_node[n][1]:
            visit(l)
        done.add(n)
        order.append(by_node[n])

    for c in range(4):
        for r in range(4):
            for k in range(8):
                visit(flat_out[8 * (4 * c + r) + k])
        marks.append(len(order))
    assert len(order) == len(luts)
    return order, marks


def emit(net, luts, flat_out, fname, S, K):
    luts, marks = cone_order(net, luts, flat_out)
    lines = []
    name = {}
    for B in range(16):
        for k in range(8):
            name[S[B][k]] = f"s[{8 * B + k}]"
            name[K[B][k]] = f"kp<{8 * B + k}>(kw)"
    outset = {n: i for i, n in enumerate(flat_out)}
    col_of = {n: i // 32 for i, n in enumerate(flat_out)}
    emitted_out = set()
    for pos, (n, leaves, tt) in enumerate(luts):
        if pos in marks:
            lines.append("    __builtin_amdgcn_sched_barrier(0);")
        name[n] = f"t{n}"
        args = [name[l] for l in leaves]
        if len(leaves) == 3:
            expr = f"lut3<0x{tt:02x}>({args[0]}, {args[1]}, {args[2]})"
        elif len(leaves) == 2:
            simple = {0x3c: f"({args[0]} ^ {args[1]})", 0xc0: f"({args[0]} & {args[1]})",
                      0xc3: f"~({args[0]} ^ {args[1]})", 0xfc: f"({args[0]} | {args[1]})"}
            expr = simple.get(tt, f"lut3<0x{tt:02x}>({args[0]}, {args[1]}, {args[1]})")
        else:
            assert tt in (0xf0, 0x0f), hex(tt)
            expr = f"~{args[0]}" if tt == 0x0f else args[0]
        lines.append(f"    const uint32_t {name[n]} = {expr};")
        if n in outset and pos + 1 in marks or pos + 1 == len(luts):
            pass
    for i, n in enumerate(flat_out):
        lines.append(f"    o[{i}] = {name[n]};")
    body = "\n".join(lines)
    return (f"// {len(luts)} LUT ops\n"
            f"__device__ __forceinline__ void {fname}(const uint32_t (&s)[128], const uint32_t (&kw)[4], uint32_t (&o)[128])\n"
            "{\n" + body + "\n}\n")


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else \
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "flashe_amd", "csrc", "bitslice", "aes_bitslice_gen.h")
    parts = ["// GENERATED by tools/bitslice/gen_bitslice.py -- do not edit.\n"
             "// Bit-sliced AES-256 round functions on 128 bit planes (plane[8*B + k] = bit k of state byte B,\n"
             "// 32 blocks per 32-bit word), as 3-input LUT (v_bitop3_b32) straight-line code.\n"
             "#pragma once\n#include <stdint.h>\n\nnamespace flashe {\nnamespace bs {\n\n"
             "template <int TT>\n__device__ __forceinline__ uint32_t lut3(uint32_t a, uint32_t b, uint32_t c)\n"
             "{\n    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);\n}\n\n"
             "// Key plane I (= 8*B + k: bit k of round-key byte B) expanded from the round key's four big-endian\n"
             "// words: all-ones iff that key bit is set.  kw is wave-uniform, so this is one scalar bit-field extract.\n"
             "template <int I>\n__device__ __forceinline__ uint32_t kp(const uint32_t (&kw)[4])\n"
             "{\n    constexpr int B = I / 8, k = I % 8, sh = 24 - 8 * (B % 4) + k;\n"
             "    return 0u - ((kw[B / 4] >> sh) & 1u);\n}\n\n"]
    stats = {}
    for final, fname in ((False, "round_main"), (True, "round_final")):
        net, S, K, flat_out = build_round(final)
        luts = map_luts(net, flat_out)
        verify(net, S, K, flat_out, luts, final)
        n_gates = sum(1 for n in range(len(net.ops)) if net.fanins(n))
        stats[fname] = (n_gates, len(luts))
        parts.append(emit(net, luts, flat_out, fname, S, K))
        parts.append("\n")
    parts.append("// ---- packed variant: 64 plane registers, two state bytes (B, B + 8) x 16 blocks per register;\n"
                 "// rk points at the round's 64 packed key planes ----\n"
                 "__device__ __forceinline__ uint32_t rot16(uint32_t x)\n{\n    return __builtin_amdgcn_alignbit(x, x, 16);\n}\n\n")
    for final, fname in ((False, "round_main_p"), (True, "round_final_p")):
        net, S, K, flat_out = build_round_packed(final)
        luts = map_luts(net, flat_out)
        verify_packed(net, S, K, flat_out, luts, final)
        text, n_lut, n_rot = emit_packed(net, luts, flat_out, fname, S, K)
        stats[fname] = (sum(1 for n in range(len(net.ops)) if net.fanins(n)), n_lut + n_rot)
        parts.append(text)
        parts.append("\n")
    parts.append("}  // namespace bs\n}  // namespace flashe\n")
    with open(out_path, "w") as f:
        f.write("".join(parts))
    for k, (g, l) in stats.items():
        print(f"{k}: {g} two-input gates -> {l} LUT3 ops ({l / 16:.1f} per byte)")


if __name__ == "__main__":
    main()
