"""dev diagnostic: non-SPLIT verify launches compared element by element with the oracle."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import falcon_r1cs_amd as frw
import frw_testlib
oracle = frw_testlib.load_oracle()
eng = frw.WitnessEngine(0)
dev = torch.device("cuda:0")
for logn, batch in [(10, 200), (10, 1600), (9, 300), (9, 2500)]:
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=5150 + batch)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.zeros((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.zeros((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nbad = 0
    seg_hist = {}
    for lo in range(0, batch, 100):
        ow, oi, ost = oracle.witness_ntt_verify(logn, sig[lo:lo+100], pk[lo:lo+100], hm[lo:lo+100], 1, threads=8)
        gw = wit[lo:lo+100].cpu().numpy().view(np.uint64)
        gi = inst[lo:lo+100].cpu().numpy().view(np.uint64)
        diff = (gw != ow).any(axis=2)
        if (gi != oi).any(): print("instance differs", logn, batch, lo)
        for i in np.nonzero(diff.any(axis=1))[0]:
            nbad += 1
            el = np.nonzero(diff[i])[0]
            segs = sorted({int(np.searchsorted(np.array(L.seg_off), e, side="right") - 1) for e in el})
            key = tuple(segs)
            seg_hist[key] = seg_hist.get(key, 0) + 1
            if nbad <= 6:
                runs = []
                start = prev = el[0]
                for e in el[1:]:
                    if e != prev + 1:
                        runs.append((int(start), int(prev))); start = e
                    prev = e
                runs.append((int(start), int(prev)))
                print("logn", logn, "batch", batch, "sig", lo + i, "n_el", len(el), "segs", segs, "runs", runs[:8], "...", len(runs))
                e0 = el[0]
                print("   first el", e0, "got", [hex(int(x)) for x in gw[i, e0]], "want", [hex(int(x)) for x in ow[i, e0]])
    print("== logn", logn, "batch", batch, "bad signatures", nbad, "by segments", seg_hist, flush=True)
