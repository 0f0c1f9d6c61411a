"""The compiler's resource report for the hot kernels, from the sources as they are (hipcc cross-compiles gfx950 without a GPU): no
scratch memory, and the waves per SIMD each kernel is written for.  Round 4's review found 28 - 112 bytes per lane of scratch in the
witness map's transform passes that the documents called removed -- nothing in the tree would have noticed.  This test does.
(The whole report, every kernel of every file: `make -C falcon-r1cs_amd/csrc resources`, committed as profiles/rNN_kernel_resources.txt.)"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_hot_kernels_have_no_scratch_and_their_waves():
    import kernel_resources as KR
    rows, bad = KR.hot_report()
    names = [k["name"] for _, k in rows]
    # the transform passes of every schedule (six, five and four stages; first, middle, fused, last), the dense and narrow bucket kernels,
    # the witness kernels of both parameter sets, the sparse products
    assert sum("ntt_pass_kernel<" in n for n in names) == len(KR.QAP_PASSES)
    assert any("msm_bucket_kernel<frw::FqField, true>" in n for n in names) and any("witness_ntt_verify_kernel<10, 1>" in n for n in names)
    assert not bad, "hot kernels with scratch or too few waves:\n" + "\n".join(KR.fmt(s, k) for s, k in bad)
    for _, k in rows:
        if "ntt_pass_kernel<" in k["name"]:
            assert k["scratch"] == 0 and k["waves"] == 3 and k["lds"] == 9216, KR.fmt("frw_qap.hip", k)


def test_the_committed_report_lists_every_hot_kernel_without_scratch():
    """profiles/r05_kernel_resources.txt is the report of the round's sources; a hot kernel missing from it, or listed with scratch,
    means the file is stale (regenerate: make -C falcon-r1cs_amd/csrc resources > profiles/r05_kernel_resources.txt)."""
    path = os.path.join(ROOT, "profiles", "r05_kernel_resources.txt")
    rows = [l.split(None, 7) for l in open(path) if l.startswith("frw_")]
    assert len(rows) > 150
    by_name = {r[7].strip(): r for r in rows}
    for needle, scratch_ok in (("frw::ntt_pass_kernel<2, 6, 0, 1, false>", 0), ("frw::ntt_pass_kernel<5, 6, 0, 1, false>", 0),
                               ("frw::ntt_pass_kernel<4, 6, 0, 0, false>", 0), ("frw::ntt_pass_kernel<5, 5, 0, 1, false>", 0),
                               ("frw::msm_bucket_kernel<frw::FqField, true>", 0), ("frw::witness_ntt_verify_kernel<10, 1>", 0),
                               ("frw::r1cs_eval_flat_kernel<true>", 12)):
        hit = [r for n, r in by_name.items() if n.startswith(needle)]
        assert hit, needle
        assert int(hit[0][4]) <= scratch_ok, hit[0]
