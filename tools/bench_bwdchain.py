"""Would the decoder backward gain from carrying both directions per launch?  One layer's GEMM + LayerNorm-backward chain:
(a) two streams, 4352 rows each (today) vs (b) one stream, 8704 rows per launch (stand-in for two-problem launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
D, F = 512, 2048
def bufs(R):
    g = lambda *s: torch.randn(*s, device=dev)
    return dict(R=R, dy=g(R, D), x=g(R, D), res=g(R, D), gam=g(D), mu=g(R), rs=g(R).abs() + 0.5, dz=g(R, D), dxd=g(R, D), dg=torch.zeros(D, device=dev), db=torch.zeros(D, device=dev),
                dh=g(R, F), h=g(R, F), w2=g(D, F), w1=g(F, D), wfc=g(D, D), wq=g(D, D), wqkv=g(3 * D, D), dqkv=g(R, 3 * D), t=g(R, D))
def ln_bwd(b):
    ops.call("sbl_add_layernorm_bwd", b["dy"].data_ptr(), b["x"].data_ptr(), b["res"].data_ptr(), b["gam"].data_ptr(), b["mu"].data_ptr(), b["rs"].data_ptr(),
             b["dz"].data_ptr(), b["dxd"].data_ptr(), b["dg"].data_ptr(), b["db"].data_ptr(), b["R"], D, 0.0, None, 0, ops._s())
def layer(b):
    R = b["R"]
    ln_bwd(b)
    ops.gemm(0, 0, R, F, D, b["dxd"], D, b["w2"], F, b["dh"], F, mask=b["h"], ldm=F)
    ops.gemm(0, 0, R, D, F, b["dh"], F, b["w1"], D, b["dz"], D, accumulate=1)
    ln_bwd(b)
    ops.gemm(0, 0, R, D, D, b["dxd"], D, b["wfc"], D, b["t"], D)
    ops.gemm(0, 0, R, D, D, b["t"], D, b["wq"], D, b["dz"], D, accumulate=1)
    ln_bwd(b)
    ops.gemm(0, 0, R, D, D, b["dxd"], D, b["wfc"], D, b["t"], D)
    ops.gemm(0, 0, R, D, 3 * D, b["dqkv"], 3 * D, b["wqkv"], D, b["dz"], D, accumulate=1)
def timed(build, n=6):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): build(s)
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / 5 / n * 1e3
b0, b1, bb = bufs(4352), bufs(4352), bufs(8704)
side = torch.cuda.Stream()
def two(s):
    side.wait_stream(s)
    with torch.cuda.stream(side):
        layer(b1)
    layer(b0)
    s.wait_stream(side)
print("one direction alone        : %7.1f us per layer" % timed(lambda s: layer(b0)))
print("two streams, 4352 rows each: %7.1f us per layer" % timed(two))
print("one stream, 8704 rows      : %7.1f us per layer" % timed(lambda s: layer(bb)))
