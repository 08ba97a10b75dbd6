"""Standalone rates of the GEMM shapes the stage-batched decoder step launches (see bench.py --dump-launches)."""
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
shapes = [(0,0,4352,2048,512),(0,0,4352,512,512),(0,0,4352,512,2048),(0,0,4352,512,1536),(0,1,2208,512,2048),(0,1,2208,512,512),
          (0,1,2208,2048,512),(0,1,2208,1536,512),(0,1,992,512,512),(0,1,992,512,2048),(0,1,672,512,512),(0,0,928,512,1024),(0,1,288,512,512)]
for ta, tb, M, N, K in shapes:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    t0 = timeit(lambda: ops.gemm(ta, tb, M, N, K, A, K, B, K if tb else N, C, N))
    fl = 2.0 * M * N * K / 1e6
    print("ta%d tb%d M=%4d N=%4d K=%4d: %6.1f us (%5.1f TF)" % (ta, tb, M, N, K, t0, fl / t0), flush=True)
