import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from sbl_for_multilingual_lip_reading_amd import detfill, dp
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
from sbl_for_multilingual_lip_reading_amd.transformer.optimizer import FusedAdam, TransformerOptimizer
import test_hip_parity as T
DEV = "cuda:0"
B, Tn, H, W, ne, nd = 2, 4, 24, 24, 1, 1
x, l2r, r2l = detfill.synthetic_batch(B, Tn, H, W, 51)
xd, ld, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV)
def train(kind):
    m = T.build_model(ne, nd).train()
    if kind == "torch":
        opt = TransformerOptimizer(torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-09), warmup_steps=2, k=0.5)
    else:
        flat = dp.FlatModel(m)
        opt = TransformerOptimizer(FusedAdam(flat, betas=(0.9, 0.98), eps=1e-09), warmup_steps=2, k=0.5)
    g1 = None
    for step in range(3):
        random.seed(100 + step)
        opt.zero_grad()
        pl, gl, pr, gr = m(xd, ld, rd)
        loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
        loss.backward()
        if step == 0:
            g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
        opt.step()
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in m.named_parameters()}, g1
runs = [train("torch"), train("fused"), train("torch"), train("fused"), train("fused")]
def cmp(a, b, tag):
    worst = (0, "")
    for n in a:
        if (n.startswith("decoder") or n.startswith("encoder")) and a[n].dim() >= 2:
            r = float((a[n] - b[n]).norm() / a[n].norm())
            if r > worst[0]: worst = (r, n)
    print(tag, "worst rel L2 %.2e %s" % worst)
cmp(runs[0][0], runs[1][0], "torch vs fused  ")
cmp(runs[0][0], runs[2][0], "torch vs torch  ")
cmp(runs[1][0], runs[3][0], "fused vs fused  ")
cmp(runs[3][0], runs[4][0], "fused vs fused 2")
cmp(runs[0][1], runs[1][1], "step-0 grads torch vs fused")
fr = max(float((runs[0][1][n] - runs[1][1][n]).norm() / runs[0][1][n].norm().clamp_min(1e-20)) for n in runs[0][1] if n.startswith("visual") and runs[0][1][n].dim() >= 2)
print("step-0 frontend matrix grads worst rel L2 %.2e" % fr)
