"""Dual (two problems per launch) vs single launches in a dependent hipGraph chain: per-kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
NREP = 100
def chain_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(NREP): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / NREP * 1e3
for M in (32, 416, 992, 2208):
    for N, K in ((512, 512), (1536, 512), (2048, 512), (512, 2048)):
        A = [torch.randn(M, K, device=dev) for _ in (0, 1)]; B = [torch.randn(N, K, device=dev) for _ in (0, 1)]
        bias = [torch.randn(N, device=dev) for _ in (0, 1)]; C = [torch.empty(M, N, device=dev) for _ in (0, 1)]
        t1 = chain_time(lambda: ops.gemm(0, 1, M, N, K, A[0], K, B[0], K, C[0], N, bias=bias[0]))
        def two():
            ops.gemm(0, 1, M, N, K, A[0], K, B[0], K, C[0], N, bias=bias[0])
            ops.gemm(0, 1, M, N, K, A[1], K, B[1], K, C[1], N, bias=bias[1])
        t2 = chain_time(two)
        t3 = chain_time(lambda: ops.gemm2(M, N, K, A[0], A[1], K, B[0], B[1], K, C[0], C[1], N, bias[0], bias[1]))
        print("M=%4d N=%4d K=%4d: single %6.1f us | two singles %6.1f us | dual %6.1f us" % (M, N, K, t1, t2, t3), flush=True)
