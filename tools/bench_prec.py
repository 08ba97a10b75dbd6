"""Accuracy and rate of the tile engine's precision modes (sbl_set_matmul_precision: 0 = fp32 MFMA, 6 / 3 / 1 = split-bf16
terms) on dense products: error against an fp64 reference (relative to sum |a||b|) and TFLOP/s, all four operand
layouts.  Usage: python tools/bench_prec.py [big|step|acc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(ta, tb, M, N, K, check=True, n=20):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    lda, ldb = (M if ta else K), (K if tb else N)
    out = []
    ref = None
    if check:
        A64, B64 = (A.t() if ta else A).double(), (B.t() if tb else B).double()
        ref = A64 @ B64
        scale = (A64.abs() @ B64.abs())
    for prec in (0, 6, 3, 1):
        ops.call("sbl_set_matmul_precision", prec)
        C.zero_()
        t = timeit(lambda: ops.gemm(ta, tb, M, N, K, A, lda, B, ldb, C, N), n)
        err = float(((C.double() - ref).abs() / scale).max()) if check else float("nan")
        out.append("p%d %7.1f us %6.1f TF err %.1e" % (prec, t, 2.0 * M * N * K / t / 1e6, err))
    ops.call("sbl_set_matmul_precision", 0)
    print("ta%d tb%d M=%5d N=%5d K=%5d | " % (ta, tb, M, N, K) + " | ".join(out), flush=True)


mode = sys.argv[1] if len(sys.argv) > 1 else "acc"
if mode == "acc":       # every layout, ragged sizes
    for ta, tb in ((0, 1), (0, 0), (1, 0), (1, 1)):
        for M, N, K in ((1000, 520, 1024), (4352, 512, 512), (640, 1536, 2048)):
            run(ta, tb, M, N, K)
elif mode == "big":
    for ta, tb in ((0, 1), (0, 0), (1, 0)):
        run(ta, tb, 8192, 4096, 4096, check=False, n=5)
        run(ta, tb, 4096, 4096, 4096, check=False, n=5)
else:
    for ta, tb, M, N, K in [(0, 0, 4352, 2048, 512), (0, 0, 4352, 512, 512), (0, 0, 4352, 512, 2048), (0, 0, 4352, 512, 1536),
                            (0, 1, 2208, 512, 2048), (0, 1, 2208, 2048, 512), (0, 1, 2208, 1536, 512), (0, 1, 992, 512, 512),
                            (1, 0, 2048, 512, 4352), (1, 0, 512, 512, 4352), (0, 1, 29696, 2048, 512)]:
        run(ta, tb, M, N, K, check=False)
