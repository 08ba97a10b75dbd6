"""Stem kernels alone (Conv3d 5x7x7 forward + statistics, BN/ReLU/pool, backward reduce, weight gradient) at the bench clip
size, per precision mode.  Usage: python tools/bench_stem.py [--tuning KNOB=VALUE]... [f32|bf16x6|bf16x3|bf16]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
N, T, H, W = 32, 29, 88, 88
while len(sys.argv) > 2 and sys.argv[1] == "--tuning":      # sbl_set_tuning measurement knobs (include/sbl_hip.h)
    k, v = sys.argv[2].split("=")
    ops.call("sbl_set_tuning", int(k), int(v))
    del sys.argv[1:3]


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

# ---- per-kernel times (each C-ABI entry point alone, 20 launches back to back)
from sbl_for_multilingual_lip_reading_amd.ops import call, _p, _s
for mode in (sys.argv[1:] or ["f32", "bf16x6"]):
    ops.set_matmul_precision(mode)
    Ho, Wo = H // 2, W // 2
    conv = torch.empty(N * T, Ho, Wo, 64, device=dev)
    stats = torch.zeros(128, device=dev, dtype=torch.float64)
    mean, invstd = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    pooled = torch.empty(N * T, Ho // 2, Wo // 2, 64, device=dev)
    argmax = torch.empty(N * T, Ho // 2, Wo // 2, 64, device=dev, dtype=torch.uint8)
    w2 = w.detach().contiguous().view(64, 245)
    gam, bet = g.detach(), b.detach()
    dpooled = torch.randn_like(pooled)
    sums = torch.zeros(128, device=dev, dtype=torch.float64)
    dw, dgam, dbet = torch.empty(64, 245, device=dev), torch.empty(64, device=dev), torch.empty(64, device=dev)
    call("sbl_stem_conv_fwd", _p(x), _p(w2), _p(conv), _p(stats), N, T, H, W, _s())
    call("sbl_bn_finalize", _p(stats), N * T * Ho * Wo, None, None, 0.1, 1e-5, _p(mean), _p(invstd), 64, None, _s())
    t = {
        "conv_fwd": timeit(lambda: call("sbl_stem_conv_fwd", _p(x), _p(w2), _p(conv), _p(stats), N, T, H, W, _s()), 20),
        "bn_relu_pool": timeit(lambda: call("sbl_stem_bn_relu_pool_fwd", _p(conv), _p(mean), _p(invstd), _p(gam), _p(bet), _p(pooled), _p(argmax), N * T, Ho, Wo, _s()), 20),
        "bwd_reduce": timeit(lambda: call("sbl_stem_bwd_reduce", _p(conv), _p(dpooled), _p(argmax), _p(mean), _p(invstd), _p(gam), _p(bet), _p(sums), N * T, Ho, Wo, _s()), 20),
        "wgrad": timeit(lambda: call("sbl_stem_wgrad", _p(x), _p(conv), _p(dpooled), _p(argmax), _p(mean), _p(invstd), _p(gam), _p(bet), _p(sums), _p(dw), _p(dgam), _p(dbet), N, T, H, W, _s()), 20),
    }
    print("%-7s per kernel (us): %s" % (mode, "  ".join("%s %.0f" % kv for kv in t.items())), flush=True)
ops.set_matmul_precision("f32")
