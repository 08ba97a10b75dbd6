"""micro-benchmark of sbl_gemm_f32 / sbl_wgrad_seg_f32 at the batched-decoder shapes (HIP events, back-to-back)"""
import os, sys, ctypes
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
Ms = [int(x) for x in os.environ.get("MS", "480,960,1440,2112,2880,4352").split(",")]
for (N, K) in [(512, 512), (1536, 512), (2048, 512), (512, 2048)]:
    for M in Ms:
        X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev); dY = torch.randn(M, N, device=dev); dX = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        t_f = timeit(lambda: ops.gemm(0, 1, M, N, K, X, K, W, K, Y, N, bias=b))
        t_dx = timeit(lambda: ops.gemm(0, 0, M, K, N, dY, N, W, K, dX, K))
        t_dw = timeit(lambda: ops.gemm(1, 0, N, K, M, dY, N, X, K, dW, K, accumulate=1, colsum=db))
        Ap = (ctypes.c_void_p * 1)(dY.data_ptr()); Bp = (ctypes.c_void_p * 1)(X.data_ptr()); rows = (ctypes.c_int * 1)(M)
        t_sw = timeit(lambda: ops.call("sbl_wgrad_seg_f32", 1, Ap, N, Bp, K, rows, N, K, dW.data_ptr(), K, db.data_ptr(), ops._s()))
        fl = 2.0 * M * N * K / 1e6
        print("M=%4d N=%4d K=%4d  fwd %6.1f us (%5.1f TF)  dX %6.1f us (%5.1f TF)  dW %6.1f us (%5.1f TF)  segdW %6.1f us (%5.1f TF)" % (
            M, N, K, t_f, fl / t_f, t_dx, fl / t_dx, t_dw, fl / t_dw, t_sw, fl / t_sw), flush=True)
