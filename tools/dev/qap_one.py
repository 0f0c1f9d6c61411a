"""One small frw_qap_witness_map_dev call (one signature): the first thing to run after touching frw_qap.hip."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import falcon_r1cs_amd as frw
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = torch.device("cuda:0"); eng = frw.WitnessEngine(0); L = frw.layout(logn)
sig, pk, hm = frw.synth_triples(logn, 1, seed=1)
d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
wit = torch.empty((1, L.num_witness, 4), dtype=torch.int64, device=dev)
inst = torch.empty((1, L.num_instance, 4), dtype=torch.int64, device=dev)
st = torch.empty(1, dtype=torch.int32, device=dev)
eng.witness_ntt_verify_dev(logn, 1, d[0], d[1], d[2], wit, inst, st, 1, 0)
torch.cuda.synchronize()
r = eng.r1cs_load(0, logn); q = eng.qap_info(r)
n, per = int(q.domain_size), int(q.workspace_bytes_per_signature)
print("n", n, "per", per, "C", q.num_constraints, "I", q.num_instance, flush=True)
ws = torch.empty(per, dtype=torch.uint8, device=dev); h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
bad = torch.empty(1, dtype=torch.int32, device=dev)
print("ws %x..%x h %x wit %x inst %x" % (ws.data_ptr(), ws.data_ptr() + per, h.data_ptr(), wit.data_ptr(), inst.data_ptr()), flush=True)
eng.qap_witness_map_dev(r, 1, wit, inst, h, ws, per, bad, 0)
torch.cuda.synchronize()
print("done; unsatisfied", bad.tolist(), "h[-1]", h[0, -1].tolist())
