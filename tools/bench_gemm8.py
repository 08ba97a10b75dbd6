"""dense products at the row counts of BOTH decoder directions together (8704 = 2 x 4352): tile experiments"""
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
for ta, tb, M, N, K in [(0,0,8704,512,512),(0,0,8704,512,2048),(0,0,8704,512,1536),(0,0,8704,2048,512),(0,0,4352,512,512),(0,0,4352,512,2048),(0,1,4416,512,512),(0,1,4416,2048,512),(0,1,4416,512,2048),(0,1,1984,512,512)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
    t0 = timeit(lambda: ops.gemm(ta, tb, M, N, K, A, K, B, K if tb else N, C, N))
    print("ta%d tb%d M=%4d N=%4d K=%4d: %6.1f us (%5.1f TF)" % (ta, tb, M, N, K, t0, 2.0 * M * N * K / 1e6 / t0), flush=True)
