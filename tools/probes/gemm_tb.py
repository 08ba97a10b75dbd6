"""dense product with the B operand row-major [K][N] (transB = 0: what the input-gradient products dX = dY W read) against
the same product with B pre-transposed to [N][K] (transB = 1: the forward's layout), alone, bf16x6."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
ops.set_matmul_precision("bf16x6")
ws = ops._workspace()
for M, N, K in [(4352, 2048, 512), (4352, 512, 2048), (4352, 512, 512), (4352, 512, 1536), (928, 512, 2048)]:
    A = torch.randn(M, K, device=dev); Bkn = torch.randn(K, N, device=dev); Bnk = Bkn.t().contiguous(); C = torch.empty(M, N, device=dev)
    def run(tb, B, acc):
        f = lambda: ops.call("sbl_gemm_f32", 0, tb, M, N, K, A.data_ptr(), K, B.data_ptr(), N if tb == 0 else K, C.data_ptr(), N, None, 0, None, 0, acc, None, ws.data_ptr(), ops.WS_BYTES, ops._s())
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 20 * 1e3
    t0, t1, t0a = run(0, Bkn, 0), run(1, Bnk, 0), run(0, Bkn, 1)
    print("M%d N%d K%d  B[K][N] %.1f us (+= %.1f)   B[N][K] %.1f us   %.0f / %.0f TF" % (M, N, K, t0, t0a, t1, 2e-6 * M * N * K / t0, 2e-6 * M * N * K / t1))
