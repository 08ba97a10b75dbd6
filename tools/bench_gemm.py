"""micro-benchmark of sbl_gemm_f32 at decoder shapes (HIP events, back-to-back launches on one stream)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
seed = torch.zeros(1, dtype=torch.int64, device=dev)
print("seed_bump (launch floor) %.1f us" % timeit(lambda: ops.call("sbl_seed_bump", seed.data_ptr(), ops._s())))
for (M, N, K) in [(32, 512, 512), (272, 512, 512), (512, 512, 512), (272, 1536, 512), (272, 2048, 512), (272, 512, 2048), (928, 512, 512)]:
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    Y = torch.empty(M, N, device=dev); dY = torch.randn(M, N, device=dev); dX = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    t_f = timeit(lambda: ops.gemm(0, 1, M, N, K, X, K, W, K, Y, N, bias=b))
    t_dx = timeit(lambda: ops.gemm(0, 0, M, K, N, dY, N, W, K, dX, K))
    t_dw = timeit(lambda: ops.gemm(1, 0, N, K, M, dY, N, X, K, dW, K, accumulate=1, colsum=db))
    fl = 2.0 * M * N * K
    print("M=%4d N=%4d K=%4d  fwd %6.1f us (%5.1f TF)  dX %6.1f us  dW %6.1f us" % (M, N, K, t_f, fl / t_f / 1e6, t_dx, t_dw))
# LN + attention at decoder sizes
for L in (1, 8, 16):
    M = 32 * L
    x = torch.randn(M, 512, device=dev); r = torch.randn(M, 512, device=dev); g = torch.ones(512, device=dev); be = torch.zeros(512, device=dev)
    y = torch.empty_like(x); mu = torch.empty(M, device=dev); rs = torch.empty(M, device=dev); dz = torch.empty_like(x); dg = torch.zeros(512, device=dev); dbb = torch.zeros(512, device=dev)
    t1 = timeit(lambda: ops.call("sbl_add_layernorm_fwd", x.data_ptr(), r.data_ptr(), g.data_ptr(), be.data_ptr(), y.data_ptr(), mu.data_ptr(), rs.data_ptr(), M, 512, 1e-5, 0.0, None, 0, ops._s()))
    t2 = timeit(lambda: ops.call("sbl_add_layernorm_bwd", y.data_ptr(), x.data_ptr(), r.data_ptr(), g.data_ptr(), mu.data_ptr(), rs.data_ptr(), dz.data_ptr(), None, dg.data_ptr(), dbb.data_ptr(), M, 512, 0.0, None, 0, ops._s()))
    q = torch.randn(32, L, 1536, device=dev); o = torch.empty(32, L, 512, device=dev); p = torch.empty(256, L, L, device=dev)
    t3 = timeit(lambda: ops.call("sbl_attention_fwd", q.data_ptr(), 1536, q[:, :, 512:].data_ptr(), 1536, q[:, :, 1024:].data_ptr(), 1536, o.data_ptr(), 512, p.data_ptr(), 0, None, 32, 8, L, L, 0.125, 0.0, None, 0, ops._s()))
    print("L=%2d  LN fwd %5.1f us  LN bwd %5.1f us  attention fwd %5.1f us" % (L, t1, t2, t3))
