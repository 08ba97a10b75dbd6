import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from sbl_for_multilingual_lip_reading_amd import ops, dp, detfill
from test_hip_parity import build_model
cnt = collections.Counter()
inner = ops.call
def call(name, *a):
    cnt[name] += 1
    return inner(name, *a)
ops.call = call
m = build_model(1, 1).train()
flat = dp.FlatModel(m)
x, l2r, r2l = detfill.synthetic_batch(4, 6, 88, 88, 3)
xd = torch.from_numpy(x).to("cuda:0")
feats = m.visual_frontend(xd.unsqueeze(4).permute(0, 4, 1, 2, 3))
feats.square().mean().backward()
torch.cuda.synchronize()
for k in ("sbl_bn_bwd_reduce", "sbl_conv2d_dgrad", "sbl_conv2d_dgrad_bnstats", "sbl_bn_bwd_apply"):
    print(k, cnt[k])
