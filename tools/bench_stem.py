"""Stem kernels alone (Conv3d 5x7x7 forward + statistics, BN/ReLU/pool, backward reduce, weight gradient) at the bench clip
size, per precision mode.  Usage: python tools/bench_stem.py [f32|bf16x6|bf16x3|bf16]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
N, T, H, W = 32, 29, 88, 88


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


x = torch.randn(N, T, H, W, device=dev, requires_grad=False)
w = (torch.randn(64, 1, 5, 7, 7, device=dev) * 0.05).requires_grad_(True)
g = torch.ones(64, device=dev, requires_grad=True); b = torch.zeros(64, device=dev, requires_grad=True)
rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
nbt = torch.zeros((), dtype=torch.long, device=dev)
for mode in (sys.argv[1:] or ["f32", "bf16x6"]):
    ops.set_matmul_precision(mode)
    y = ops.StemFn.apply(x, w, g, b, rm, rv, True, 0.1, 1e-5, nbt)
    dy = torch.randn_like(y)
    tf = timeit(lambda: ops.StemFn.apply(x, w, g, b, rm, rv, True, 0.1, 1e-5, nbt))
    def fb():
        yy = ops.StemFn.apply(x, w, g, b, rm, rv, True, 0.1, 1e-5, nbt)
        yy.backward(dy)
    tfb = timeit(fb)
    print("%-7s stem forward %.0f us, forward+backward %.0f us" % (mode, tf, tfb), flush=True)
ops.set_matmul_precision("f32")
