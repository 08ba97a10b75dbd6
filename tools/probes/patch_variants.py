"""patch-kernel variants (sbl_set_tuning knob 5: 0 per-tap gathers, 1 padded 64-channel rows, 2 swizzled 32-channel rows):
agreement of forward / input gradient with variant 0 and the time of each alone.
usage: patch_variants.py [H C]   (default 22 64 = layer 1; 11 128 = layer 2, two images per tile)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
ops.set_matmul_precision("bf16x6")
NIMG, H, C = 928, 22, 64
if len(sys.argv) > 2:
    H, C = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(NIMG, H, H, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
w_ohwi = torch.empty(C, 3, 3, C, device=dev); w_dg = torch.empty(C, 3, 3, C, device=dev)
ops.call("sbl_conv_weight_pack", w.data_ptr(), w_ohwi.data_ptr(), w_dg.data_ptr(), C, C, 3, 3, None, 0, ops._s())
dy = torch.randn(NIMG, H, H, C, device=dev)
def run(k):
    ops.call("sbl_set_tuning", 5, k)
    y = torch.empty(NIMG, H, H, C, device=dev); stats = torch.zeros(2 * C, device=dev, dtype=torch.float64)
    dx = torch.empty_like(x)
    f = lambda: ops.call("sbl_conv2d_fwd", x.data_ptr(), w_ohwi.data_ptr(), y.data_ptr(), stats.data_ptr(), 0, NIMG, H, H, C, C, 3, 3, 1, 1, ops._workspace().data_ptr(), ops.WS_BYTES, ops._s())
    g = lambda: ops.call("sbl_conv2d_dgrad", dy.data_ptr(), w_dg.data_ptr(), dx.data_ptr(), NIMG, H, H, C, C, 3, 3, 1, 1, ops._workspace().data_ptr(), ops.WS_BYTES, ops._s())
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 20 * 1e3
    tf, tg = t(f), t(g)
    stats.zero_(); f(); g(); torch.cuda.synchronize()
    return y, dx, stats.clone(), tf, tg
y0, dx0, s0, *_ = run(0)
for k in (0, 1, 2):
    y, dx, st, tf, tg = run(k)
    print("knob5=%d  fwd %.0f us  dgrad %.0f us   max|dy| %.2e  max|ddx| %.2e  stats rel %.2e" % (k, tf, tg, float((y - y0).abs().max()), float((dx - dx0).abs().max()), float(((st - s0).abs() / s0.abs().clamp_min(1)).max())))
