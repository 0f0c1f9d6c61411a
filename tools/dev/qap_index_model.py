"""Python model of the tile geometry of frw_qap.hip (which working indices a tile holds per pass, the bit-reversed first
pass that reads the products in constraint order, where the fused element-wise steps sit) with exact integers, checked
against oracle/qap.py on a small domain.  The butterflies here still take one twiddle per stage from the full table
(the kernel's predecessor); the four-step form the kernel uses now -- 64-th roots inside a pass, one per-index factor
after it -- is modelled in qap_fourstep_model.py and qap_radix8_model.py.   python tools/dev/qap_index_model.py [L=12]"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import qap

P = qap.P
FIRST, DIT_SH0, DIT, DIF, DIF_SH0 = range(5)


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def run_pass(mode, L, sh, T, x, tw, load=None, store=None):
    """x: dict array (list) in working order, modified in place; load(tileid, lin)->value for FIRST; store(g, v)."""
    n = 1 << L
    R, elems = 1 << T, (1 << T) * 16
    dif = mode in (DIF, DIF_SH0)
    sh0 = mode in (DIT_SH0, DIF_SH0)
    lowj = mode in (DIT, DIF)
    touched = set()
    for tileid in range(n >> (T + 4)):
        lowmid = high = 0
        if mode == FIRST:
            high = brev(tileid, L - 10) << 6
        elif sh0:
            high = tileid * elems
        else:
            lowmid = (tileid & ((1 << (sh - 4)) - 1)) << 4
            high = (tileid >> (sh - 4)) << (sh + T)

        def widx(lin):
            if mode == FIRST:
                return (lin >> 4) | high | (brev(lin & 15, 4) << (L - 4))
            if sh0:
                return high + lin
            return high | ((lin >> 4) << sh) | lowmid | (lin & 15)
        tile = [None] * elems
        if mode == FIRST:
            for lin in range(elems):
                ihi, c = lin >> 4, lin & 15
                i = (ihi << (L - 6)) | (tileid << 4) | c
                assert 0 <= i < n
                tile[brev(ihi, 6) * 16 + c] = load(i)
        else:
            for lin in range(elems):
                g = widx(lin)
                assert 0 <= g < n
                tile[lin] = load(g) if load else x[g]
        for ti in range(T):
            t = T - ti if dif else ti + 1
            s, hr = sh + t, 1 << (t - 1)
            for k in range(elems // 2):
                b, c = (k & (R // 2 - 1), k >> (T - 1)) if sh0 else (k >> 4, k & 15)
                r_lo = b & (hr - 1)
                r = ((b >> (t - 1)) << t) | r_lo
                s0 = c * R + r if sh0 else r * 16 + c
                s1 = s0 + (hr if sh0 else hr * 16)
                assert s1 < elems
                j = ((r_lo << sh) | lowmid | c) if lowj else r_lo
                e = j << (L - s)
                assert e < n // 2, (mode, s, j, e)
                u, v = tile[s0], tile[s1]
                if dif:
                    tile[s0], tile[s1] = (u + v) % P, (u - v) * tw[e] % P
                else:
                    v = v * tw[e] % P
                    tile[s0], tile[s1] = (u + v) % P, (u - v) % P
        for lin in range(elems):
            g = widx(lin)
            assert 0 <= g < n and g not in touched
            touched.add(g)
            x[g] = store(g, tile[lin]) if store else tile[lin]
    assert len(touched) == n


def passes(L):
    return [(sh, min(6, L - sh)) for sh in range(0, L, 6)]


def witness_map_model(az, bz, cz, num_inputs, z):
    nc = len(az)
    d = qap.Domain(nc + num_inputs)
    L, n = d.log_size, d.size
    tw_f = [pow(d.group_gen, k, P) for k in range(n // 2)]
    tw_i = [pow(d.group_gen_inv, k, P) for k in range(n // 2)]
    s_in = [pow(qap.GENERATOR, k, P) * d.size_inv % P for k in range(n)]
    zinv = pow((pow(qap.GENERATOR, n, P) - 1) % P, P - 2, P)
    s_out = [pow(d.generator_inv, k, P) * d.size_inv % P * zinv % P for k in range(n)]
    arrays = []
    for which, src in enumerate((az, bz, cz)):
        x = [None] * n

        def load(i, src=src, which=which):
            if i < nc:
                return src[i]
            if which == 0 and i - nc < num_inputs:
                return z[i - nc] % P
            return 0
        ps = passes(L)
        for idx, (sh, T) in enumerate(ps):                       # inverse, decimation in time
            last = idx == len(ps) - 1
            st = (lambda g, v: v * s_in[g] % P) if last else None
            run_pass(FIRST if sh == 0 else DIT, L, sh, T, x, tw_i, load if sh == 0 else None, st)
        for idx, (sh, T) in enumerate(reversed(ps)):             # forward, decimation in frequency
            run_pass(DIF_SH0 if sh == 0 else DIF, L, sh, T, x, tw_f)
        arrays.append(x)
    a, b, c = arrays
    h = [None] * n
    ps = passes(L)
    for idx, (sh, T) in enumerate(ps):
        last = idx == len(ps) - 1
        st = (lambda g, v: v * s_out[g] % P) if last else None
        ld = (lambda g: (a[g] * b[g] - c[g]) % P) if sh == 0 else None
        run_pass(DIT_SH0 if sh == 0 else DIT, L, sh, T, h, tw_i, ld, st)
    return h


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = random.Random(L)
    num_inputs = 37
    nc = (1 << L) - num_inputs - 5
    az = [rng.randrange(P) for _ in range(nc)]
    bz = [rng.randrange(P) for _ in range(nc)]
    cz = [rng.randrange(P) for _ in range(nc)]
    z = [1] + [rng.randrange(P) for _ in range(num_inputs - 1)]
    want = qap.witness_map_from_products(az, bz, cz, num_inputs, z)
    got = witness_map_model(az, bz, cz, num_inputs, z)
    assert got == want
    print("index model == oracle for L = %d" % L)


if __name__ == "__main__":
    main()
