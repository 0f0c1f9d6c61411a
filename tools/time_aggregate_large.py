"""Phase times of ONE proof for a large aggregate (default: BASELINE configs[4]'s 1,024 mixed statements, the 2^27 domain):
witnesses, handle (transform tables on the device), key (device setup), proof (repeated), verification.  Prints JSON.
    python tools/time_aggregate_large.py [statements=1024] [proofs=3] [1024: Falcon-1024 only | mixed] [auto | tables | bare]"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import falcon_r1cs_amd as frw  # noqa: E402

SEED = 0x46414C434F4E31
R_FR = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    only10 = len(sys.argv) > 3 and sys.argv[3] == "1024"
    rng = random.Random(SEED)
    logns = [10] * total if only10 else [rng.choice([9, 10]) for _ in range(total)]
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    s0 = torch.cuda.current_stream().cuda_stream
    out = {"statements": total, "falcon512": logns.count(9), "falcon1024": logns.count(10)}
    t0 = time.perf_counter()
    batches = {}
    for g in (9, 10):
        cnt = logns.count(g)
        if not cnt:
            continue
        L = frw.layout(g)
        sig, pk, hm = frw.synth_triples(g, cnt, SEED, (1 << 43) + (g << 20))
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        wit = torch.empty((cnt, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((cnt, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(cnt, dtype=torch.int32, device=dev)
        eng.witness_ntt_verify_dev(g, cnt, d[0], d[1], d[2], wit, inst, st, frw.ENC_MONTGOMERY, s0)
        torch.cuda.synchronize()
        assert not st.any()
        batches[g] = (wit, inst)
    out["witnesses_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    handle = eng.r1cs_load_aggregate(logns)
    out["handle_s"] = round(time.perf_counter() - t0, 3)
    info = eng.r1cs_info(handle)
    ni, nw, nc, n = int(info.num_instance), int(info.num_witness), int(info.num_constraints), 1 << int(info.log_domain_size)
    out.update(num_instance=ni, num_witness=nw, num_constraints=nc, log_domain=int(info.log_domain_size))
    d_wit = torch.empty((1, nw, 4), dtype=torch.int64, device=dev)
    d_inst = torch.empty((1, ni, 4), dtype=torch.int64, device=dev)
    b9, b10 = batches.get(9, (None, None)), batches.get(10, (None, None))
    eng.aggregate_assign_dev(handle, b9[0], b9[1], b10[0], b10[1], d_wit, d_inst, s0)
    torch.cuda.synchronize()
    toxic = [rng.randrange(2, R_FR) for _ in range(5)]
    t0 = time.perf_counter()
    mode = {"auto": frw.KEY_AUTO, "tables": frw.KEY_TABLES, "bare": frw.KEY_BARE}[sys.argv[4] if len(sys.argv) > 4 else "auto"]
    key, vk = eng.groth16_setup_r1cs(handle, *toxic, mode=mode)
    out["key_s"] = round(time.perf_counter() - t0, 3)
    pi = eng.groth16_pk_info(key)
    out["key_mode"], out["key_bytes"] = int(pi.mode), int(pi.key_bytes)
    ws_bytes = eng.groth16_workspace_bytes(key, handle, 1)
    out["workspace_bytes"] = ws_bytes
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    proof = torch.empty((1, 48), dtype=torch.int64, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    lim = lambda ks: np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    rs = np.stack([lim([rng.randrange(R_FR), rng.randrange(R_FR)])])
    run = lambda: eng.groth16_prove_dev(key, handle, 1, d_wit, d_inst, rs, proof, ws, ws_bytes, bad, s0)
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    out["first_proof_s"] = round(time.perf_counter() - t0, 3)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    assert bad.tolist() == [0]
    # the order of a bucket's entries is whatever the sorts' atomics made it; the proof must not depend on it: every repetition's bytes
    first = proof.clone()
    same = 0
    for _ in range(int(os.environ.get("FRW_TOOL_SOAK", "0"))):
        proof.zero_()
        run()
        torch.cuda.synchronize()
        assert torch.equal(proof, first), "the proof's bytes changed between two runs on the same inputs"
        same += 1
    if same:
        out["soak_identical_proofs"] = same + 1
    out["ms_per_proof"] = round(ms, 2)
    out["signatures_per_s"] = round(total / (ms * 1e-3), 1)
    if int(pi.mode) == frw.KEY_BARE:
        z = torch.cat([d_inst[0], d_wit[0], torch.zeros((3, 4), dtype=torch.int64, device=dev)]).contiguous()
        c = eng.diag_groth16_side_counts(key, z, ws, ws_bytes, s0)
        out["witness_side"] = {"rows_of_b_queries_holding_a_point": c[0], "of_rows": ni + nw + 3, "digits_all": c[1], "ones_all": c[2], "digits_b": c[3], "ones_b": c[4]}
        del z
    # the witness map on its own
    q = eng.qap_info(handle)
    h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
    eng.qap_witness_map_dev(handle, 1, d_wit, d_inst, h, ws, ws_bytes, None, s0)
    torch.cuda.synchronize()
    e0.record(stream)
    eng.qap_witness_map_dev(handle, 1, d_wit, d_inst, h, ws, ws_bytes, None, s0)
    e1.record(stream)
    torch.cuda.synchronize()
    out["witness_map_ms"] = round(e0.elapsed_time(e1), 2)
    hq = eng.groth16_pk_query(key, 0)
    hacc = torch.empty((1, 12), dtype=torch.int64, device=dev)
    eng.groth16_msm_h_dev(hq, 1, h, n, hacc, ws, ws_bytes, s0)
    torch.cuda.synchronize()
    e0.record(stream)
    eng.groth16_msm_h_dev(hq, 1, h, n, hacc, ws, ws_bytes, s0)
    e1.record(stream)
    torch.cuda.synchronize()
    out["h_query_sum_ms"] = round(e0.elapsed_time(e1), 2)
    out["peak_allocated_bytes"] = int(torch.cuda.max_memory_allocated())
    free_b, total_b = torch.cuda.mem_get_info()
    out["device_bytes_in_use"] = int(total_b - free_b)
    eng.groth16_pk_free(key)
    del ws, h
    t0 = time.perf_counter()
    ver = frw.Groth16Verifier(vk, points_are_checked=True)
    out["vk_load_s"] = round(time.perf_counter() - t0, 3)
    inst_h = d_inst.cpu().numpy().view(np.uint64)
    proof_h = proof.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    ok = ver.verify(inst_h, proof_h).tolist()
    out["verify_s"] = round(time.perf_counter() - t0, 3)
    assert ok == [1]
    ver.close()
    eng.r1cs_free(handle)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
