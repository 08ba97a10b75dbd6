"""dense GEMMs at the trunk convolutions' implicit-GEMM shapes (what the tile engine does without the im2col gather)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, N, K in [("layer1", 449152, 64, 576), ("layer2", 112288, 128, 1152), ("layer3", 33408, 256, 2304), ("layer4", 8352, 512, 4608)]:
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); Y = torch.empty(M, N, device=dev)
    dY = torch.randn(M, N, device=dev); dW = torch.zeros(N, K, device=dev)
    t_f = timeit(lambda: ops.gemm(0, 1, M, N, K, X, K, W, K, Y, N))
    t_w = timeit(lambda: ops.gemm(1, 0, N, K, M, dY, N, X, K, dW, K, accumulate=1))
    fl = 2.0 * M * N * K / 1e6
    print("%s M=%6d N=%3d K=%4d  NT %7.1f us (%5.1f TF)   TN(dW) %7.1f us (%5.1f TF)" % (name, M, N, K, t_f, fl / t_f, t_w, fl / t_w), flush=True)
