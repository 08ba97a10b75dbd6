"""per-CU efficiency probe: shapes whose tile count is an exact multiple of 256 CUs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, N, K) in [(4096, 2048, 512), (4096, 2048, 4096), (8192, 4096, 512), (8192, 4096, 4096), (2048, 1024, 512), (2048, 1024, 4096), (2048, 2048, 512), (1024, 1024, 512), (1024,1024,4096)]:
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    Y = torch.empty(M, N, device=dev); dY = torch.randn(M, N, device=dev); dX = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev)
    t_f = timeit(lambda: ops.gemm(0, 1, M, N, K, X, K, W, K, Y, N, bias=b))
    t_dx = timeit(lambda: ops.gemm(0, 0, M, K, N, dY, N, W, K, dX, K))
    t_dw = timeit(lambda: ops.gemm(1, 0, N, K, M, dY, N, X, K, dW, K, accumulate=1))
    fl = 2.0 * M * N * K / 1e6
    print("M=%4d N=%4d K=%4d  fwd %7.1f us (%5.1f TF)  dX %7.1f us (%5.1f TF)  dW %7.1f us (%5.1f TF)" % (M, N, K, t_f, fl / t_f, t_dx, fl / t_dx, t_dw, fl / t_dw), flush=True)
