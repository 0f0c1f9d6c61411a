"""A/B: does the latency-bound front of a proof call (the four witness-side sums, the scalar multiples, the witness map) hide
behind ANOTHER call's 2^18-point bucket kernel?  The same 64 proofs as one call on one stream, and as two / four calls of 32 / 16
on streams of their own (separate workspaces; one key, whose side streams the calls share).
usage: python tools/ab_groth16_overlap.py [logn=10] [batch=64] [reps=5]"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=1)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    r1cs = eng.r1cs_load(0, logn)
    rng = random.Random(3)
    key, vk = eng.groth16_setup(0, logn, *(rng.randrange(2, R) for _ in range(5)))
    lim = lambda ks: np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    rs = np.stack([lim([rng.randrange(R), rng.randrange(R)]) for _ in range(batch)])
    ref = None
    for parts in (1, 2, 4, 1):
        per = batch // parts
        streams = [torch.cuda.Stream() for _ in range(parts)]
        ws_bytes = eng.groth16_workspace_bytes(key, r1cs, per)
        ws = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(parts)]
        proofs = torch.zeros((batch, 48), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

        def run():
            for p in range(parts):
                lo = p * per
                eng.groth16_prove_dev(key, r1cs, per, wit[lo:lo + per], inst[lo:lo + per], rs[lo:lo + per], proofs[lo:lo + per], ws[p], ws_bytes,
                                      None, streams[p].cuda_stream)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        for _ in range(reps):
            run()
        for s in streams:
            e = torch.cuda.Event()
            e.record(s)
            torch.cuda.current_stream().wait_event(e)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if ref is None:
            ref = proofs.clone()
        assert torch.equal(proofs, ref), "proofs differ between the arrangements"
        print("Falcon-%d, %d proofs as %d call(s) of %d on %d stream(s): %.2f ms = %.1f proofs/s" % (
            1 << logn, batch, parts, per, parts, ms, batch / ms * 1e3), flush=True)
        del ws


if __name__ == "__main__":
    main()
