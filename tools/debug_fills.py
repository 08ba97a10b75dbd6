import os, sys, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from sbl_for_multilingual_lip_reading_amd import detfill, dp
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
import test_hip_parity as T
DEV = "cuda:0"
B, Tn, H, W = 16, 4, 24, 24
x, l2r, r2l = detfill.synthetic_batch(B, Tn, H, W, 41)
xd, ld, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV)
m = T.build_model(1, 2).train()
flat = dp.FlatModel(m)
def step():
    random.seed(13)
    flat.zero_grad()
    pl, gl, pr, gr = m(xd, ld, rd)
    loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
    return loss
step().backward()
loss = step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    loss.backward()
torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::zeros_like", "aten::new_zeros", "aten::add_", "aten::add", "aten::copy_"):
        # parent chain
        p = e.cpu_parent
        names = []
        while p is not None and len(names) < 4:
            names.append(p.name); p = p.cpu_parent
        c[(e.name, tuple(names), str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(25):
    print(v, k)
