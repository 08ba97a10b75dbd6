"""debug: which of {flat accumulation, two streams} perturbs r2l gradients"""
import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from sbl_for_multilingual_lip_reading_amd import detfill, dp
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
from test_hip_parity import build_model
DEV = "cuda:0"
B, T, H, W, ne, nd = 3, 4, 24, 24, 1, 2
x, l2r, r2l = detfill.synthetic_batch(B, T, H, W, 33)
xd, ld, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV)

def run(flat, two):
    m = build_model(ne, nd).train()
    m.decoder.two_streams = two
    f = dp.FlatModel(m) if flat else None
    if f: f.zero_grad()
    random.seed(9)
    pl, gl, pr, gr = m(xd, ld, rd)
    loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

ref = run(False, False)
for flat, two in ((True, False), (False, True), (True, True)):
    g = run(flat, two)
    worst = {}
    for n in ref:
        if n.endswith("w_ks.bias"):
            continue
        e = float((g[n] - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-7))
        key = ".".join(n.split(".")[:2])
        if e > worst.get(key, (0, ""))[0]:
            worst[key] = (e, n)
    print("flat=%s two=%s" % (flat, two), {k: ("%.1e" % v[0], v[1]) for k, v in worst.items() if v[0] > 1e-3})
