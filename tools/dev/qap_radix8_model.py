"""Python model of frw_qap.hip's register rounds (round_low / round_high: which rows a thread holds, which pairs it
combines, which 64-th root it uses) against the plain 2^T-point sub-transform of qap_fourstep_model.py, for T = 6 and 5,
both directions.   python tools/dev/qap_radix8_model.py"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qap_fourstep_model import sub_transform
from oracle import qap

P = qap.P


def dit(x, i, j, w):
    v = x[j] * w % P
    x[i], x[j] = (x[i] + v) % P, (x[i] - v) % P


def dif(x, i, j, w):
    u, v = x[i], x[j]
    x[i], x[j] = (u + v) % P, (u - v) * w % P


def round_low(x, roots, is_dif):
    w8, w16, w24 = roots[8], roots[16], roots[24]
    if not is_dif:
        for a, b in ((0, 1), (2, 3), (4, 5), (6, 7)): dit(x, a, b, 1)
        dit(x, 0, 2, 1); dit(x, 1, 3, w16); dit(x, 4, 6, 1); dit(x, 5, 7, w16)
        dit(x, 0, 4, 1); dit(x, 1, 5, w8); dit(x, 2, 6, w16); dit(x, 3, 7, w24)
    else:
        dif(x, 0, 4, 1); dif(x, 1, 5, w8); dif(x, 2, 6, w16); dif(x, 3, 7, w24)
        dif(x, 0, 2, 1); dif(x, 1, 3, w16); dif(x, 4, 6, 1); dif(x, 5, 7, w16)
        for a, b in ((0, 1), (2, 3), (4, 5), (6, 7)): dif(x, a, b, 1)


def round_high(x, roots, g, T, is_dif):
    op = dif if is_dif else dit
    if T == 6:
        s4 = [lambda: [op(x, a, a + 1, roots[g << 2]) for a in (0, 2, 4, 6)]]
        s5 = [lambda: [op(x, 0, 2, roots[g << 1]), op(x, 1, 3, roots[(g + 8) << 1]), op(x, 4, 6, roots[g << 1]), op(x, 5, 7, roots[(g + 8) << 1])]]
        s6 = [lambda: [op(x, e, e + 4, roots[g + 8 * e]) for e in range(4)]]
        order = (s6 + s5 + s4) if is_dif else (s4 + s5 + s6)
    else:
        wa, wb = roots[g << 2], roots[(g + 4) << 2]
        wa0, wa1, wb0, wb1 = roots[g << 1], roots[(g + 8) << 1], roots[(g + 4) << 1], roots[(g + 12) << 1]
        s4 = [lambda: [op(x, 0, 1, wa), op(x, 2, 3, wa), op(x, 4, 5, wb), op(x, 6, 7, wb)]]
        s5 = [lambda: [op(x, 0, 2, wa0), op(x, 1, 3, wa1), op(x, 4, 6, wb0), op(x, 5, 7, wb1)]]
        order = (s5 + s4) if is_dif else (s4 + s5)
    for f in order:
        f()


def tile_transform(vals, T, roots, is_dif):
    R = 1 << T
    groups = R // 8
    row_low = lambda g, e: g * 8 + e
    row_high = (lambda g, e: e * 8 + g) if T == 6 else (lambda g, e: (e & 3) * 8 + g + 4 * (e >> 2))
    cur = list(vals)
    first, second = (row_high, row_low) if is_dif else (row_low, row_high)
    for rnd, rows in enumerate((first, second)):
        nxt = [None] * R
        for g in range(groups):
            x = [cur[rows(g, e)] for e in range(8)]
            is_low = rows is row_low
            if is_low:
                round_low(x, roots, is_dif)
            else:
                round_high(x, roots, g, T, is_dif)
            for e in range(8):
                assert nxt[rows(g, e)] is None
                nxt[rows(g, e)] = x[e]
        assert None not in nxt
        cur = nxt
    return cur


def main():
    rng = random.Random(1)
    n = 1 << 18
    w = qap.Domain(n).group_gen
    roots = [pow(w, k * (n >> 6), P) for k in range(32)]
    for T in (6, 5):
        for is_dif in (False, True):
            vals = [rng.randrange(P) for _ in range(1 << T)]
            assert tile_transform(vals, T, roots, is_dif) == sub_transform(vals, T, roots, 18, is_dif), (T, is_dif)
    print("register rounds == plain sub-transform for T = 6, 5, both directions")


if __name__ == "__main__":
    main()
