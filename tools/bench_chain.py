"""dependent chain of decoder-shaped GEMMs with cold (cycled) weights, replayed from a hipGraph: per-kernel cost
as the decoder sees it (no host launch overhead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
NW = 200
def chain(M, K, N, label, kind):
    Ws = [torch.randn(N, K, device=dev) * 0.02 for _ in range(NW)]
    b = torch.zeros(N, device=dev)
    xs = [torch.randn(M, K, device=dev), torch.empty(M, N, device=dev)]
    dW = torch.zeros(N, K, device=dev)
    side = torch.cuda.Stream()
    def run():
        for i in range(NW):
            if kind == "fwd":      # y = x W^T (+b); square shapes chain x->y->x
                ops.gemm(0, 1, M, N, K, xs[i & 1] if K == N else xs[0], K, Ws[i], K, xs[(i + 1) & 1] if K == N else xs[1], N, bias=b)
            elif kind == "dx":
                ops.gemm(0, 0, M, K, N, xs[1] if K != N else xs[i & 1], N, Ws[i], K, xs[0] if K != N else xs[(i + 1) & 1], K)
            else:
                ops.gemm(1, 0, N, K, M, xs[1] if K != N else xs[0], N, xs[0], K, Ws[i], K, accumulate=1, colsum=b)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 5 / NW * 1e3
    print("%-4s M=%4d N=%4d K=%4d : %6.2f us per GEMM in a dependent chain (%.1f TF)" % (kind, M, N, K, us, 2.0 * M * N * K / us / 1e6))
import sys as _sys
shapes = [(1536, 512, 512), (1536, 512, 2048), (1536, 2048, 512), (3072, 512, 512), (768, 512, 512)]
for (M, K, N) in shapes:
    chain(M, K, N, "", "fwd")
for (M, K, N) in shapes[:3]:
    chain(M, K, N, "", "dx")
    chain(M, K, N, "", "dw")
