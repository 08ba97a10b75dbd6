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
for M, N, K in [(131072, 128, 1152), (98304, 128, 1152), (65536, 256, 2304), (32768, 256, 2304), (32768, 512, 4608), (16384, 512, 4608), (131072, 128, 4096), (65536, 256, 512)]:
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); Y = torch.empty(M, N, device=dev)
    t_f = timeit(lambda: ops.gemm(0, 1, M, N, K, X, K, W, K, Y, N))
    fl = 2.0 * M * N * K / 1e6
    print("M=%6d N=%3d K=%4d  NT %7.1f us (%5.1f TF)  tiles128=%d" % (M, N, K, t_f, fl / t_f, (M // 128) * ((N + 127) // 128)), flush=True)
