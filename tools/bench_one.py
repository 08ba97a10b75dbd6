"""run one sbl_gemm_f32 shape N times (for rocprofv3 --pmc):  bench_one.py M N K mode[fwd|dx|dw] reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); mode = sys.argv[4]; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda:0"
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
Y = torch.empty(M, N, device=dev); dY = torch.randn(M, N, device=dev); dX = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev)
for _ in range(reps):
    if mode == "fwd": ops.gemm(0, 1, M, N, K, X, K, W, K, Y, N, bias=b)
    elif mode == "dx": ops.gemm(0, 0, M, K, N, dY, N, W, K, dX, K)
    else: ops.gemm(1, 0, N, K, M, dY, N, X, K, dW, K, accumulate=1)
torch.cuda.synchronize()
