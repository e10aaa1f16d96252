#!/usr/bin/env python3
"""Where kPhiExp in flashe_amd/csrc/mt19937.hip comes from: the characteristic polynomial phi of MT19937's state transition, found with
Berlekamp-Massey on one output bit of the raw word sequence (it has degree 19937 and 135 terms), and a check of the jump relation the
device code relies on:  with g = x^J mod phi,   x[1 + J + w] = XOR over k with g_k = 1 of x[1 + k + w].

    python tools/mt_jump_poly.py          # prints the 135 exponents (takes a few seconds)
"""
import numpy as np

N, M, DEG = 624, 397, 19937


def twist(mt):
    new = mt.copy()
    for k in range(N):
        y = (int(new[k]) & 0x80000000) | (int(new[(k + 1) % N]) & 0x7fffffff)
        new[k] = int(new[(k + M) % N]) ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
    return new


def raw_sequence(key, nwords):
    out = [np.array(key, dtype=np.uint32)]
    while sum(len(o) for o in out) < nwords:
        out.append(twist(out[-1]))
    return np.concatenate(out)[:nwords]


def berlekamp_massey(bits):
    """Connection polynomial C (as an int, bit i = C_i) and linear complexity L of a GF(2) sequence."""
    sint = 0
    for i, b in enumerate(bits):
        sint |= b << i
    C, B, L, m = 1, 1, 0, 1
    for n in range(len(bits)):
        if n >= L:
            window = (sint >> (n - L)) & ((1 << (L + 1)) - 1)                 # bit j = s[n - L + j]
            c_rev = int(bin(C | (1 << (L + 1)))[3:][::-1], 2)                 # C reversed over L + 1 bits
            d = bin(window & c_rev).count("1") & 1
        else:
            d = 0
            for i in range(n + 1):
                d ^= (C >> i) & 1 & bits[n - i]
        if d == 0:
            m += 1
        elif 2 * L <= n:
            C, B, L, m = C ^ (B << m), C, n + 1 - L, 1
        else:
            C ^= B << m
            m += 1
    return C, L


def main():
    np.random.seed(5489)
    key = np.random.get_state()[1].copy()
    seq = raw_sequence(key, 2 * DEG + 2 * N)
    C, L = berlekamp_massey([int(v) & 1 for v in seq[1:2 * DEG + 3]])        # y_i = x[i + 1]: a coordinate sequence of the state
    assert L == DEG
    phi = int(bin(C | (1 << (L + 1)))[3:][::-1], 2)                          # phi_k = C_(L - k)
    exps = [i for i in range(phi.bit_length()) if (phi >> i) & 1]
    print(len(exps), "terms:", exps)

    def polymod(p):
        while p.bit_length() > DEG:
            p ^= phi << (p.bit_length() - 1 - DEG)
        return p

    def mulmod(a, b):
        acc = 0
        while a:
            if a & 1:
                acc ^= b
            b <<= 1
            a >>= 1
        return polymod(acc)

    J, g, base, e = 20000, 1, 2, 20000
    while e:
        if e & 1:
            g = mulmod(g, base)
        base = mulmod(base, base)
        e >>= 1
    seq = raw_sequence(key, J + DEG + 3 * N)
    for w in (0, 1, 623):
        acc, k, gg = 0, 0, g
        while gg:
            if gg & 1:
                acc ^= int(seq[1 + k + w])
            gg >>= 1
            k += 1
        assert acc == int(seq[1 + J + w])
    print("jump relation holds for J =", J)


if __name__ == "__main__":
    main()
