"""R1CS matrix export: frw_r1cs_export writes A, B, C with every symbolic LC inlined.

Checked against the ORACLE's independent inlining (oracle/ark_sim.py::to_matrices on the restated gadgets): identical
matrices, entry by entry, for both circuits at N = 512; and the oracle's witness satisfies the exported matrices."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import falcon_gadgets as G

P = G.P_BLS12_381_FR


def read_r1cs(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"FRWR1CS1"
    ni, nw, nc, *nnz = np.frombuffer(raw, dtype=np.uint64, count=6, offset=8).tolist()
    off = 8 + 48
    mats = []
    for k in range(3):
        ptr = np.frombuffer(raw, dtype=np.uint64, count=nc + 1, offset=off); off += 8 * (nc + 1)
        col = np.frombuffer(raw, dtype=np.uint32, count=nnz[k], offset=off); off += 4 * nnz[k]
        val = np.frombuffer(raw, dtype=np.uint64, count=4 * nnz[k], offset=off).reshape(-1, 4); off += 32 * nnz[k]
        assert int(ptr[-1]) == nnz[k]
        mats.append((ptr, col, val))
    assert off == len(raw)
    return ni, nw, nc, mats


def export(circuit, logn, path):
    import falcon_r1cs_amd as frw
    cnt = (C.c_uint64 * 6)()
    assert frw.load_library().frw_r1cs_export(circuit, logn, str(path).encode(), cnt) == 0
    return list(cnt)


@pytest.mark.parametrize("circuit", [0, 1])
def test_exported_matrices_equal_oracle_inlining(tmp_path, circuit):
    logn = 9
    rng = random.Random(13)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    flow = G.run_reference_flow_dual if circuit else G.run_reference_flow
    cs = flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    want = cs.to_matrices()
    path = tmp_path / "c.r1cs"
    cnt = export(circuit, logn, path)
    ni, nw, nc, mats = read_r1cs(path)
    assert (ni, nw, nc) == (cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()) == tuple(cnt[:3])
    for (ptr, col, val), rows in zip(mats, want):
        assert int(ptr[-1]) == sum(len(r) for r in rows)
        flat_cols = np.fromiter((c for r in rows for c, _ in r), dtype=np.uint32)
        assert np.array_equal(col, flat_cols)
        assert np.array_equal(np.diff(ptr.astype(np.int64)), np.fromiter((len(r) for r in rows), dtype=np.int64))
        flat_vals = b"".join(v.to_bytes(32, "little") for r in rows for _, v in r)
        assert val.tobytes() == flat_vals


@pytest.mark.parametrize("circuit,logn", [(0, 9), (0, 10), (1, 9)])
def test_oracle_witness_satisfies_exported_matrices(tmp_path, oracle, circuit, logn):
    rng = random.Random(17 + logn)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    fn = oracle.witness_dual_ntt_verify if circuit else oracle.witness_ntt_verify
    wit, inst, st = fn(logn, sig, pk, hm, 0)                       # canonical encoding
    assert st[0] == 0
    to_int = lambda rows: [sum(int(x) << (64 * i) for i, x in enumerate(r)) for r in rows]
    z = to_int(inst[0]) + to_int(wit[0])
    path = tmp_path / "c.r1cs"
    export(circuit, logn, path)
    ni, nw, nc, mats = read_r1cs(path)
    assert ni + nw == len(z)
    prods = []
    for ptr, col, val in mats:
        vals = [int(a) | int(b) << 64 | int(c) << 128 | int(d) << 192 for a, b, c, d in val.tolist()]
        cols = col.tolist()
        p_ = ptr.tolist()
        prods.append([sum(vals[k] * z[cols[k]] for k in range(p_[i], p_[i + 1])) % P for i in range(nc)])
    az, bz, cz = prods
    assert all((a * b - c) % P == 0 for a, b, c in zip(az, bz, cz))
    # and a corrupted witness does not
    z[ni + 5] += 1
    prods = []
    for ptr, col, val in mats:
        vals = [int(a) | int(b) << 64 | int(c) << 128 | int(d) << 192 for a, b, c, d in val.tolist()]
        cols = col.tolist()
        p_ = ptr.tolist()
        prods.append([sum(vals[k] * z[cols[k]] for k in range(p_[i], p_[i + 1])) % P for i in range(nc)])
    assert any((a * b - c) % P for a, b, c in zip(*prods))
