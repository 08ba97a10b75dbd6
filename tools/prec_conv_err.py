"""Error of the trunk convolutions (fwd / dgrad / wgrad) against fp64 under each precision mode, relative to the result's
max: tells an arithmetic difference (x3 vs x6 terms) from an indexing bug."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from sbl_for_multilingual_lip_reading_amd import ops
DEV = "cuda:0"
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
torch.manual_seed(0)
shapes = [(15, 10, 6, 64, 64, 3, 1), (15, 10, 6, 64, 128, 3, 2), (15, 10, 6, 64, 128, 1, 2), (15, 5, 3, 128, 128, 3, 1), (15, 5, 3, 128, 256, 3, 2),
          (15, 3, 2, 256, 256, 3, 1), (15, 3, 2, 256, 512, 3, 2), (15, 3, 2, 256, 512, 1, 2), (15, 2, 1, 512, 512, 3, 1), (928, 3, 3, 512, 512, 3, 1), (928, 6, 6, 256, 256, 3, 1),
          (200, 22, 22, 64, 64, 3, 1), (200, 11, 11, 128, 128, 3, 1), (200, 22, 22, 64, 128, 3, 2)]
for (NIMG, H, W, Cin, Cout, k, stride) in shapes:
    pad = 1 if k == 3 else 0
    x = torch.randn(NIMG, Cin, H, W).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k) * 0.1).requires_grad_(True)
    y = F.conv2d(x.double(), w.double(), None, stride, pad)
    dy = torch.randn(y.shape)
    y.backward(dy.double())
    Ho, Wo = y.shape[2:]
    xd, wd, dyd = nhwc(x.detach()).to(DEV), w.detach().to(DEV), nhwc(dy).to(DEV)
    w_ohwi = torch.empty(Cout, k, k, Cin, device=DEV); w_dg = torch.empty(Cin, k, k, Cout, device=DEV)
    ops.call("sbl_conv_weight_pack", wd.data_ptr(), w_ohwi.data_ptr(), w_dg.data_ptr(), Cout, Cin, k, k, None, 0, ops._s())
    ws = ops._workspace()
    line = "%4dx%2dx%2d c%3d->%3d k%d s%d |" % (NIMG, H, W, Cin, Cout, k, stride)
    for mode in ("f32", "bf16x6", "bf16x3"):
        ops.set_matmul_precision(mode)
        yd = torch.empty(NIMG, Ho, Wo, Cout, device=DEV)
        ops.call("sbl_conv2d_fwd", xd.data_ptr(), w_ohwi.data_ptr(), yd.data_ptr(), None, 0, NIMG, H, W, Cin, Cout, k, k, stride, pad, ws.data_ptr(), ops.WS_BYTES, ops._s())
        dxd = torch.empty_like(xd)
        ops.call("sbl_conv2d_dgrad", dyd.data_ptr(), w_dg.data_ptr(), dxd.data_ptr(), NIMG, H, W, Cin, Cout, k, k, stride, pad, ws.data_ptr(), ops.WS_BYTES, ops._s())
        dwd = torch.empty(Cout, k, k, Cin, device=DEV)
        ops.call("sbl_conv2d_wgrad", xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), NIMG, H, W, Cin, Cout, k, k, stride, pad, 0, ops._s())
        def rel(a, b): return float((a.cpu().double() - b).abs().max() / b.abs().max())
        line += " %s fwd %.1e dg %.1e wg %.1e |" % (mode, rel(yd, nhwc(y.detach())), rel(dxd, nhwc(x.grad)), rel(dwd, w.grad.permute(0, 2, 3, 1)))
    print(line, flush=True)
ops.set_matmul_precision("f32")
