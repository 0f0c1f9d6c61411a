"""Python model of the four-step form of frw_qap.hip's transforms: every pass = plain 2^T-point sub-transforms with the
2^T-th roots only (radix-8 rounds in registers on the device), followed by ONE multiplication per element by a per-index
factor (the twist for the next pass / this pass, or the scale factor of the witness map).  Exact integers, checked
against oracle/qap.py.   python tools/dev/qap_fourstep_model.py [L=13]"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import qap

P = qap.P


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def schedule(L):
    """frw_device.h qap_pass_schedule restated: stages per pass, lowest bits first; pass 0 has six, the others four to six."""
    if L < 14 or L > 30:
        return None
    k = (L + 5) // 6
    deficit, t = 6 * k - L, [6] * k
    for _ in range(2):
        for i in range(k - 1, 0, -1):
            if deficit:
                t[i] -= 1
                deficit -= 1
    return None if deficit else t


def passes(L):
    t = schedule(L)
    if t is None:                                        # below the device's range: the model alone, six bits at a time
        return [(sh, min(6, L - sh)) for sh in range(0, L, 6)]
    return [(sum(t[:k]), t[k]) for k in range(len(t))]


def twist_exponent(idx, L, sh, T):
    """Exponent of w by which the element at working index idx is multiplied before (DIT) / after (DIF) the pass on bits
    [sh, sh + T): low bits times the bit-reversed row, scaled to the 2^(sh+T)-th root."""
    low, r = idx & ((1 << sh) - 1), (idx >> sh) & ((1 << T) - 1)
    return (low * brev(r, T)) << (L - sh - T)


def sub_transform(vals, T, root64, L, dif):
    """plain size-2^T transform on a list (row index = position): DIT bit-reversed in -> natural out, or DIF natural -> bitrev."""
    R = 1 << T
    a = list(vals)
    for ti in range(T):
        t = T - ti if dif else ti + 1
        hr = 1 << (t - 1)
        for b in range(R // 2):
            r_lo = b & (hr - 1)
            r = ((b >> (t - 1)) << t) | r_lo
            w = root64[r_lo << (6 - t)]                  # 64-th roots table, index = r_lo 2^(6 - t)
            u, v = a[r], a[r + hr]
            if dif:
                a[r], a[r + hr] = (u + v) % P, (u - v) * w % P
            else:
                v = v * w % P
                a[r], a[r + hr] = (u + v) % P, (u - v) % P
    return a


def run_pass(x, L, sh, T, root64, dif, post):
    """in place on the working array; post(idx, value) applied on the way out"""
    n = 1 << L
    out = [None] * n
    for base in range(n):
        if (base >> sh) & ((1 << T) - 1):
            continue                                     # base = index with row bits zero
        rows = [base | (r << sh) for r in range(1 << T)]
        res = sub_transform([x[i] for i in rows], T, root64, L, dif)
        for i, v in zip(rows, res):
            out[i] = post(i, v) if post else v
    x[:] = out


def witness_map_model(az, bz, cz, num_inputs, z):
    nc = len(az)
    d = qap.Domain(nc + num_inputs)
    L, n = d.log_size, d.size
    w, wi = d.group_gen, d.group_gen_inv
    r64f = [pow(w, k * (n >> 6), P) for k in range(32)]
    r64i = [pow(wi, k * (n >> 6), P) for k in range(32)]
    s_in = [pow(qap.GENERATOR, k, P) * d.size_inv % P for k in range(n)]
    zinv = pow((pow(qap.GENERATOR, n, P) - 1) % P, P - 2, P)
    s_out = [pow(d.generator_inv, k, P) * d.size_inv % P * zinv % P for k in range(n)]
    ps = passes(L)

    def dit(x, final_scale):
        for k, (sh, T) in enumerate(ps):
            if k + 1 < len(ps):
                nsh, nT = ps[k + 1]
                post = lambda i, v, nsh=nsh, nT=nT: v * pow(wi, twist_exponent(i, L, nsh, nT), P) % P
            else:
                post = lambda i, v: v * final_scale[i] % P
            run_pass(x, L, sh, T, r64i, False, post)

    def dif(x):
        for sh, T in reversed(ps):
            post = (lambda i, v, sh=sh, T=T: v * pow(w, twist_exponent(i, L, sh, T), P) % P) if sh else None
            run_pass(x, L, sh, T, r64f, True, post)
    arrays = []
    for which, src in enumerate((az, bz, cz)):
        nat = list(src) + [0] * (n - nc)
        if which == 0:
            nat[nc:nc + num_inputs] = [v % P for v in z[:num_inputs]]
        x = [nat[brev(pos, L)] for pos in range(n)]      # the first pass reads through the bit-reversed tile
        dit(x, s_in)
        dif(x)
        arrays.append(x)
    a, b, c = arrays
    h = [(a[i] * b[i] - c[i]) % P for i in range(n)]
    dit(h, s_out)
    return h


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    rng = random.Random(L)
    num_inputs = 37
    nc = (1 << L) - num_inputs - 5
    az, bz, cz = ([rng.randrange(P) for _ in range(nc)] for _ in range(3))
    z = [1] + [rng.randrange(P) for _ in range(num_inputs - 1)]
    want = qap.witness_map_from_products(az, bz, cz, num_inputs, z)
    got = witness_map_model(az, bz, cz, num_inputs, z)
    assert got == want
    print("four-step model == oracle for L = %d, passes %s" % (L, passes(L)))


if __name__ == "__main__":
    main()
