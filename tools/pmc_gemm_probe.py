"""One dense product of the tile engine, repeated, for a rocprofv3 --pmc pass (tools/pmc_gemm_probe.sh).
Usage: python tools/pmc_gemm_probe.py ta tb M N K prec [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
ta, tb, M, N, K, prec = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = "cuda:0"
A = torch.randn((K, M) if ta else (M, K), device=dev)
B = torch.randn((N, K) if tb else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
ops.call("sbl_set_matmul_precision", prec)
for _ in range(reps):
    ops.gemm(ta, tb, M, N, K, A, M if ta else K, B, K if tb else N, C, N)
torch.cuda.synchronize()
