import sys, random, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_hip_parity as tp
from sbl_for_multilingual_lip_reading_amd import dp, ops
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
ops.set_matmul_precision("bf16x6")
DEV = tp.DEV
B, T, H, W, ne, nd = 16, 4, 24, 24, 1, 2
x, l2r, r2l = tp.detfill.synthetic_batch(B, T, H, W, 33)
xd, ld, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV)
def run(m):
    random.seed(9)
    pl, gl, pr, gr = m(xd, ld, rd)
    loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
    loss.backward()
    return pl.detach().clone(), loss.item()
for trial in range(3):
    m1 = tp.build_model(ne, nd).train(); m1.decoder.two_streams = False
    pl1, loss1 = run(m1)
    g1 = {n: p.grad.clone() for n, p in m1.named_parameters()}
    m2 = tp.build_model(ne, nd).train(); flat = dp.FlatModel(m2); flat.zero_grad()
    pl2, loss2 = run(m2); torch.cuda.synchronize()
    print("trial", trial, "fwd diff", float((pl2 - pl1).abs().max()), "loss diff", abs(loss1 - loss2))
    for n, p in m2.named_parameters():
        ref = g1[n]; d = (p.grad - ref).abs()
        rel = float(d.max()) / (float(ref.abs().max()) + 1e-30)
        if rel > 2e-4 and not n.startswith("visual_frontend"):
            msg = "%s rel %.2e" % (n, rel)
            if d.dim() == 2:
                rows = (d.max(dim=1).values > 0.1 * d.max()).nonzero().flatten().tolist()
                cols = (d.max(dim=0).values > 0.1 * d.max()).nonzero().flatten().tolist()
                msg += " rows>10%%max: %d %s cols: %d" % (len(rows), rows[:6], len(cols))
            print("   ", msg)
