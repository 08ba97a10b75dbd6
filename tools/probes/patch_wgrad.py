"""patch-resident weight gradient (sbl_set_tuning knob 9) against the implicit-GEMM weight gradients: time of each alone and
agreement, at the trunk's 3x3 / stride-1 shapes.  usage: patch_wgrad.py [H C [knob]] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
ops.set_matmul_precision("bf16x6")
NIMG = 928
shapes = [(22, 64, 100), (11, 128, 100), (6, 256, 30), (3, 512, 9)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)]
for H, C, knob in shapes:
    x = torch.randn(NIMG, H, H, C, device=dev); dy = torch.randn(NIMG, H, H, C, device=dev)
    def run(k):
        ops.call("sbl_set_tuning", 9, k)
        dw = torch.zeros(C, 3, 3, C, device=dev)
        f = lambda z: ops.call("sbl_conv2d_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), NIMG, H, H, C, C, 3, 3, 1, 1, z, ops._s())
        for _ in range(3): f(1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f(1)
        b.record(); torch.cuda.synchronize()
        f(0); torch.cuda.synchronize()
        return dw.clone(), a.elapsed_time(b) / 20 * 1e3
    d0, t0 = run(0)
    d1, t1 = run(knob)
    ops.call("sbl_set_tuning", 9, 30)
    print("%2dx%-2d c%-3d  gather %.0f us   patch %.0f us   rel diff %.2e" % (H, H, C, t0, t1, float((d1 - d0).abs().max() / d0.abs().max())))
