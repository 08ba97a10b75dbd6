"""per-kernel time inside a hipGraph chain (no host launch floor): N back-to-back launches of one kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
NREP = 200
def chain_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(NREP): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / NREP * 1e3
seed = torch.zeros(1, dtype=torch.int64, device=dev)
print("empty-ish kernel (seed_bump): %.2f us" % chain_time(lambda: ops.call("sbl_seed_bump", seed.data_ptr(), ops._s())))
for M in (96, 416, 960):
    x = torch.randn(M, 512, device=dev); r = torch.randn(M, 512, device=dev); g_ = torch.ones(512, device=dev); be = torch.zeros(512, device=dev)
    y = torch.empty_like(x); mu = torch.empty(M, device=dev); rs = torch.empty(M, device=dev)
    t = chain_time(lambda: ops.call("sbl_add_layernorm_fwd", x.data_ptr(), r.data_ptr(), g_.data_ptr(), be.data_ptr(), y.data_ptr(), mu.data_ptr(), rs.data_ptr(), M, 512, 1e-5, 0.0, None, 0, ops._s()))
    out = ["M=%4d LN fwd %.2f us |" % (M, t)]
    for K in (64, 128, 256, 512, 1024, 2048):
        X = torch.randn(M, K, device=dev); W = torch.randn(512, K, device=dev); b = torch.randn(512, device=dev); Y = torch.empty(M, 512, device=dev)
        t = chain_time(lambda: ops.gemm(0, 1, M, 512, K, X, K, W, K, Y, 512, bias=b))
        out.append("K=%d: %.2f" % (K, t))
    print(" ".join(out), flush=True)
