"""numerical debugging: run the same train step under several decoder schedules and compare gradients"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from sbl_for_multilingual_lip_reading_amd import detfill
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
import test_hip_parity as T
DEV = "cuda:0"
B, Tn, H, W, ne, nd = 3, 4, 24, 24, 1, 2
x, l2r, r2l = detfill.synthetic_batch(B, Tn, H, W, 41)
xd, ld, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV)
def run(batched, two):
    m = T.build_model(ne, nd).train()
    m.decoder.batch_teacher_runs = batched
    m.decoder.two_streams = two
    random.seed(13)
    pl, gl, pr, gr = m(xd, ld, rd)
    loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}
ref = run(False, False)
import collections
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for it in range(N):
    g = run(True, True)
    off = collections.defaultdict(list)
    for n in g:
        if n.endswith('w_ks.bias'): continue
        d = float((g[n] - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-20))
        if d > 1e-4:
            off[n.split(".")[0]].append((n, "%.1e" % d))
    if off:
        bad += 1
        print(it, {k: (len(v), v[:6]) for k, v in off.items()}, flush=True)
print("anomalous runs: %d of %d" % (bad, N))
