"""Finite-difference probe of the decoder backward with dropout on: per parameter, both decoder paths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from sbl_for_multilingual_lip_reading_amd import ops, dp
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
from sbl_for_multilingual_lip_reading_amd import detfill
from test_hip_parity import build_model
DEV = "cuda:0"
B, T, H, W, ne, nd = 16, 4, 24, 24, 1, 2
x, l2r, r2l = detfill.synthetic_batch(B, T, H, W, 73)
xd, ld, rd = torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV)
for batched in (True, False):
  for pdrop in (0.0, 0.1):
    m = build_model(ne, nd).train()
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = pdrop
    m.decoder.coins_host = [False] * 16
    m.decoder.batched_backward = batched
    flat = dp.FlatModel(m)
    st = ops.dropout_state(torch.device(DEV))
    with torch.no_grad():
        feats = m.visual_frontend(xd.unsqueeze(4).permute(0, 4, 1, 2, 3))
    lengths = [T] * B
    def loss_fn():
        st._offset = 0
        enc, *_ = m.encoder(feats, lengths)
        pl, gl, pr, gr = m.decoder(ld, rd, enc, lengths)
        return 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
    flat.zero_grad()
    loss_fn().backward()
    ops.join_side_streams(); torch.cuda.synchronize()
    grad = flat.flat_grad.clone()
    names = {id(p): n for n, p in m.named_parameters()}
    gen = torch.Generator(DEV).manual_seed(5)
    print("batched", batched, "p", pdrop)
    for p_, off, n in flat.slots:
        nm = names[id(p_)]
        if not nm.startswith("decoder."): continue
        if ".bias" in nm and "layer_norm" not in nm and "w_qs" not in nm: continue
        v = torch.zeros_like(flat.flat_param)
        v[off:off + p_.numel()] = torch.randn(p_.numel(), device=DEV, generator=gen)
        ana = float((grad.double() * v.double()).sum())
        eps = 1e-3 / max(1.0, float(v.norm()) / 30)
        base = flat.flat_param.clone()
        vals = []
        for sgn in (1.0, -1.0):
            flat.flat_param.copy_(base + sgn * eps * v)
            vals.append(float(loss_fn().detach().double()))
        flat.flat_param.copy_(base)
        num = (vals[0] - vals[1]) / (2 * eps)
        flag = "  <<<<" if abs(num - ana) > 0.03 * abs(ana) + 2e-3 else ""
        print(f"  {nm:55s} num {num:10.4f} ana {ana:10.4f}{flag}")
