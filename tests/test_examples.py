"""examples/constraint_counts.cpp -- counterpart of the reference's examples/constraint_counts.rs (BASELINE configs[0])."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "build", "constraint_counts")


@pytest.fixture(scope="module")
def example_bin():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", BIN, os.path.join(ROOT, "examples", "constraint_counts.cpp"),
                           "-L" + os.path.join(ROOT, "falcon-r1cs_amd"), "-lfrw", "-Wl,-rpath,$ORIGIN/../../../falcon-r1cs_amd"])
    return BIN


def _table(out):
    rows = {}
    for line in out.splitlines():
        cells = [c.strip() for c in line.strip("|").split("|")]
        if len(cells) == 4 and cells[1].isdigit():
            rows[cells[0]] = tuple(int(c) for c in cells[1:])
    return rows


def test_count_table_matches_reference_readme(example_bin):
    """README.md:43-44,54-55 (the dual row is printed by the reference but not published)."""
    for logn, want in ((10, {"ntt conversion": (0, 29696, 30720), "verify with ntt": (2049, 156724, 162870)}),
                       (9, {"ntt conversion": (0, 14848, 15360), "verify with ntt": (1025, 78386, 81460)})):
        out = subprocess.run([example_bin, str(logn)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        rows = _table(out.stdout)
        for name, counts in want.items():
            assert rows[name] == counts


@pytest.mark.gpu
def test_example_satisfied_on_engine(example_bin):
    for logn in (9, 10):
        out = subprocess.run([example_bin, str(logn)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.count("satisfied") >= 2 and "NOT satisfied" not in out.stdout, out.stdout
        assert "HIP engine" in out.stdout
