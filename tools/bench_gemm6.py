"""NN input-gradient GEMMs with a 512-wide output and K = 1536 / 2048 (dX of the QKV and FFN-w1 projections), plain and +="""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for K in (1536, 2048):
    for M in (416, 544, 640, 800, 960, 1280, 1440, 2112):
        dY = torch.randn(M, K, device=dev); W = torch.randn(K, 512, device=dev); dX = torch.zeros(M, 512, device=dev)
        t0 = timeit(lambda: ops.gemm(0, 0, M, 512, K, dY, K, W, 512, dX, 512))
        t1 = timeit(lambda: ops.gemm(0, 0, M, 512, K, dY, K, W, 512, dX, 512, accumulate=1))
        fl = 2.0 * M * 512 * K / 1e6
        print("M=%4d K=%4d -> 512:  plain %6.1f us (%5.1f TF)   += %6.1f us (%5.1f TF)" % (M, K, t0, fl / t0, t1, fl / t1), flush=True)
