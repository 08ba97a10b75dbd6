"""How far each precision mode lands from the CPU oracle on the unfixtured e2e case of tests/test_hip_parity.py
(test_e2e_matches_oracle_other_seed): worst gradient deviation relative to the gradient's max, logits, features."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import sbl_oracle as O
from sbl_for_multilingual_lip_reading_amd import detfill, ops
from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
import test_hip_parity as T
DEV = "cuda:0"
for (B, Tt, H, W, ne, nd, seed, cs) in [(3, 5, 40, 24, 1, 2, 21, 5), (2, 8, 24, 24, 2, 2, 3, 7), (8, 6, 32, 32, 1, 1, 4, 2)]:
    x, l2r, r2l = detfill.synthetic_batch(B, Tt, H, W, seed)
    sd = O.make_state_dict(ne, nd, requires_grad=True)
    random.seed(cs)
    coins = O.draw_coins()
    ref = O.transformer_forward(sd, torch.from_numpy(x), torch.from_numpy(l2r), torch.from_numpy(r2l), coins, ne, nd)
    O.train_step_loss(ref).backward()
    for mode in ("f32", "bf16x6", "bf16x3"):
        ops.set_matmul_precision(mode)
        m = T.build_model(ne, nd).train()
        random.seed(cs)
        pl, gl, pr, gr = m(torch.from_numpy(x).to(DEV), torch.from_numpy(l2r).to(DEV), torch.from_numpy(r2l).to(DEV))
        loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
        loss.backward()
        worst_t, worst_f, nt, nf = 0.0, 0.0, "", ""
        for n, p in m.named_parameters():
            r = sd[n].grad
            e = float((p.grad.cpu() - r).abs().max()) / (float(r.abs().max()) + 1e-3)     # (+1e-3: bound of the tests' absolute term)
            if n.startswith("visual"):
                if e > worst_f: worst_f, nf = e, n
            elif e > worst_t: worst_t, nt = e, n
        print("B%d T%d %dx%d %d+%d %-7s dlogit %.2e  transformer grads %.2e (%s)  frontend grads %.2e (%s)" % (
            B, Tt, H, W, ne, nd, mode, float((pl.cpu() - ref["pred_l2r"]).abs().max()), worst_t, nt, worst_f, nf), flush=True)
ops.set_matmul_precision("f32")
