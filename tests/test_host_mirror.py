"""The C++ host mirror (falcon-r1cs_amd/csrc/host/frw_host.hpp): an R1CS emitter written against the reference's
gadget definitions, independent of both the witness kernels and the oracle's closed form.

CPU: its structure reproduces README.md:41-56; a witness produced by the ORACLE satisfies the system it emits (and a
corrupted one does not).  GPU: the reference's own unit tests, re-written in C++ (tests/cpp/test_host_mirror.cpp),
with every witness value coming from the HIP engine."""
import os
import random
import subprocess

import numpy as np
import pytest

import frw_testlib as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "build")
BIN = os.path.join(BUILD, "test_host_mirror")


@pytest.fixture(scope="module")
def mirror_bin():
    src = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    hdr = os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "host", "frw_host.hpp")
    lib = os.path.join(ROOT, "falcon-r1cs_amd", "libfrw.so")
    assert os.path.exists(lib), "libfrw.so not built; run __graft_entry__.build()"
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(p) for p in (src, hdr, lib)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", BIN, src,
                               "-L" + os.path.dirname(lib), "-lfrw", "-Wl,-rpath,$ORIGIN/../../../falcon-r1cs_amd"])
    return BIN


def test_structure_counts_match_readme(mirror_bin):
    out = subprocess.run([mirror_bin, "structure"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "156724" in out.stdout and "162870" in out.stdout and "78386" in out.stdout and "81460" in out.stdout
    assert "29696" in out.stdout and "30720" in out.stdout and "14848" in out.stdout and "15360" in out.stdout


@pytest.mark.parametrize("logn", [9, 10])
def test_oracle_witness_satisfies_cpp_constraint_system(mirror_bin, oracle, tmp_path, logn):
    rng = random.Random(31 + logn)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, 1)
    assert st[0] == 0
    files = {}
    for name, arr in (("sig", sig), ("pk", pk), ("hm", hm), ("wit", wit), ("inst", inst)):
        files[name] = str(tmp_path / (name + ".bin"))
        arr.tofile(files[name])
    cmd = [mirror_bin, "check", str(logn)] + [files[k] for k in ("sig", "pk", "hm", "wit", "inst")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "satisfied" in out.stdout and "UNSATISFIED" not in out.stdout, out.stdout + out.stderr
    # flip one bit of one boolean witness deep inside S4: the system must notice
    n = 1 << logn
    bad = wit.copy()
    idx = 58 * n + 29 * 17 + 5
    bad[0, idx] = wit[0, idx] ^ np.uint64(1)
    bad.tofile(files["wit"])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "UNSATISFIED" in out.stdout
    # a wrong quotient t (first mod_q block of S3) as well
    bad = wit.copy()
    bad[0, 29 * n, 0] += np.uint64(1)
    bad.tofile(files["wit"])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2


@pytest.mark.parametrize("logn", [9, 10])
def test_oracle_dual_witness_satisfies_cpp_constraint_system(mirror_bin, oracle, tmp_path, logn):
    """FalconDualNTTVerificationCircuit (falcon_dual_ntt.rs): C++ emitter vs the oracle's closed-form witness."""
    rng = random.Random(41 + logn)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    wit, inst, st = oracle.witness_dual_ntt_verify(logn, sig, pk, hm, 1)
    assert st[0] == 0
    files = {}
    for name, arr in (("sig", sig), ("pk", pk), ("hm", hm), ("wit", wit), ("inst", inst)):
        files[name] = str(tmp_path / (name + ".bin"))
        arr.tofile(files[name])
    cmd = [mirror_bin, "check-dual", str(logn)] + [files[k] for k in ("sig", "pk", "hm", "wit", "inst")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "UNSATISFIED" not in out.stdout, out.stdout + out.stderr
    n = 1 << logn
    bad = wit.copy()
    bad[0, 3 * n + 1, 0] ^= np.uint64(1)            # the is_zero multiplier of the signature's DualPolyVar
    bad.tofile(files["wit"])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode in (0, 2)                # multiplier is unconstrained when acc == 0: (0 - acc) * m = 0 holds
    bad = wit.copy()
    bad[0, 2 * n + 7, 0] = np.uint64(5)             # a pos*neg product that is not the product
    bad.tofile(files["wit"])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2


@pytest.mark.gpu
def test_reference_unit_tests_on_the_engine(mirror_bin):
    """test_mod_q, test_add_mod, test_range_proof_*, test_ntt_mul_circuit, test_ntt_verification_r1cs -- values from HIP."""
    out = subprocess.run([mirror_bin, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all ok" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
