"""Do two dependent chains of small kernels on two streams (captured as parallel graph branches) overlap?
One chain of N launches vs two chains of N side by side vs one chain of 2N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
N = 200
def timed(build):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build(s, 2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        build(s, N)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 * 1e3
side = torch.cuda.Stream()
for M in (32, 416, 2208):
    bufs = []
    for d in range(2):
        X = torch.randn(M, 512, device=dev); W = torch.randn(512, 512, device=dev); b = torch.randn(512, device=dev)
        Y = torch.empty(M, 512, device=dev); r = torch.randn(M, 512, device=dev); g_ = torch.ones(512, device=dev)
        y2 = torch.empty_like(Y); mu = torch.empty(M, device=dev); rs = torch.empty(M, device=dev)
        bufs.append((X, W, b, Y, r, g_, y2, mu, rs))
    def chain(d, n):
        X, W, b, Y, r, g_, y2, mu, rs = bufs[d]
        for i in range(n):
            if i % 2 == 0:
                ops.gemm(0, 1, M, 512, 512, X, 512, W, 512, Y, 512, bias=b)
            else:
                ops.call("sbl_add_layernorm_fwd", Y.data_ptr(), r.data_ptr(), g_.data_ptr(), b.data_ptr(), y2.data_ptr(), mu.data_ptr(), rs.data_ptr(), M, 512, 1e-5, 0.0, None, 0, ops._s())
    def one(s, n): chain(0, n)
    def double_len(s, n): chain(0, n); chain(1, n)
    def two(s, n):
        side.wait_stream(s)
        with torch.cuda.stream(side):
            chain(1, n)
        chain(0, n)
        s.wait_stream(side)
    t1, t2, t3 = timed(one), timed(two), timed(double_len)
    print("M=%4d: one chain of %d: %.0f us (%.2f us/kernel) | two chains side by side: %.0f us | one chain of %d: %.0f us" % (M, N, t1, t1 / N, t2, 2 * N, t3), flush=True)
