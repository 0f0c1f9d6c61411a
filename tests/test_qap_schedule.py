"""The pass schedule of the witness map's transforms for every domain the device takes (frw_device.h qap_pass_schedule, restated
in tools/dev/qap_fourstep_model.py): pass 0 has six radix-2 stages, the others four to six, together log n; and the four-step
form with THAT schedule -- plain 2^T-point sub-transforms with the 64-th roots only, one per-index twist between passes, the twist
of working index i before the pass on bits [sh, sh + T) being w^((i mod 2^sh) bitrev_T((i >> sh) mod 2^T) 2^(L - sh - T)) -- is
ark-groth16's witness_map exactly (oracle/qap.py), on the smallest domain with a four-stage pass.  (2^15 = 6 + 5 + 4 and
2^19 = 6 + 5 + 4 + 4 were run by hand: 7 s and 3 min; the device's transforms for 2^18 .. 2^22 are held to the oracle in
tests/test_gpu_aggregate.py.)"""
import importlib.util
import os
import random
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("qap_fourstep_model", os.path.join(ROOT, "tools", "dev", "qap_fourstep_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_schedule_covers_every_domain_and_matches_the_header():
    m = _model()
    assert m.schedule(13) is None and m.schedule(31) is None
    for L in range(14, 31):
        t = m.schedule(L)
        assert sum(t) == L and t[0] == 6 and all(4 <= x <= 6 for x in t) and len(t) == (L + 5) // 6, (L, t)
    assert m.schedule(17) == [6, 6, 5] and m.schedule(18) == [6, 6, 6]          # the per-signature circuits: as in rounds 2-3
    assert m.schedule(19) == [6, 5, 4, 4] and m.schedule(20) == [6, 5, 5, 4] and m.schedule(22) == [6, 6, 5, 5] and m.schedule(24) == [6] * 4
    # the C++ header states the same rule (the arithmetic is restated, the text is compared: both files must change together)
    hdr = open(os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_device.h")).read()
    body = hdr[hdr.index("inline int qap_pass_schedule"):]
    body = body[:body.index("struct QapDev")]
    assert re.search(r"const int k = \(L \+ 5\) / 6;", body) and re.search(r"int deficit = 6 \* k - L;", body)
    assert re.search(r"for \(int round = 0; round < 2 && deficit; round\+\+\)\s*\n\s*for \(int i = k - 1; i >= 1 && deficit; i--\) \{ t\[i\]--; deficit--; \}", body)
    assert "QAP_MIN_LOG_N = 14, QAP_MAX_LOG_N = 30" in hdr


def test_four_step_form_with_a_four_stage_pass_is_the_witness_map():
    m = _model()
    L = 14
    rng = random.Random(L)
    num_inputs = 37
    nc = (1 << L) - num_inputs - 5
    az, bz, cz = ([rng.randrange(m.P) for _ in range(nc)] for _ in range(3))
    z = [1] + [rng.randrange(m.P) for _ in range(num_inputs - 1)]
    assert m.passes(L) == [(0, 6), (6, 4), (10, 4)]
    assert m.witness_map_model(az, bz, cz, num_inputs, z) == m.qap.witness_map_from_products(az, bz, cz, num_inputs, z)
