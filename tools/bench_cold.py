"""Small-M products with COLD weights: a dependent chain that cycles through NW different weight matrices (more than the
L2s hold), the way a decoder run walks its layers.  Per-kernel time for single / dual launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
NW = int(os.environ.get("NW", "48"))
def chain_time(fn, nrep):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(nrep): fn(i)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / nrep * 1e3
for M in (32, 96, 416):
    for N, K in ((512, 512), (1536, 512), (2048, 512), (512, 2048)):
        W = [torch.randn(N, K, device=dev) for _ in range(NW)]
        A = [torch.randn(M, K, device=dev) for _ in (0, 1)]; bias = torch.randn(N, device=dev); C = [torch.empty(M, N, device=dev) for _ in (0, 1)]
        warm = chain_time(lambda i: ops.gemm(0, 1, M, N, K, A[0], K, W[0], K, C[0], N, bias=bias), 96)
        cold = chain_time(lambda i: ops.gemm(0, 1, M, N, K, A[0], K, W[i % NW], K, C[0], N, bias=bias), 96)
        cold2 = chain_time(lambda i: ops.gemm2(M, N, K, A[0], A[1], K, W[(2 * i) % NW], W[(2 * i + 1) % NW], K, C[0], C[1], N, bias, bias), 96)
        print("M=%4d N=%4d K=%4d (%.1f MB/weight): warm %5.1f us | cold single %5.1f us | cold dual %5.1f us" % (M, N, K, N * K * 4 / 1e6, warm, cold, cold2), flush=True)
