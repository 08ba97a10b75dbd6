"""isolated timings of the decoder's small kernels at stage sizes (B=32)"""
import os, sys, ctypes
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
print("launch floor (seed_bump) %.1f us" % timeit(lambda: ops.call("sbl_seed_bump", seed.data_ptr(), ops._s())))
B, H = 32, 8
for segL in [(6, 7), (13,), (2, 3, 4, 5, 6)]:
    M = B * sum(segL)
    x = torch.randn(M, 512, device=dev); r = torch.randn(M, 512, device=dev); g = torch.ones(512, device=dev); be = torch.zeros(512, device=dev)
    y = torch.empty_like(x); mu = torch.empty(M, device=dev); rs = torch.empty(M, device=dev); dz = torch.empty_like(x); dxd = torch.empty_like(x)
    dg = torch.zeros(512, device=dev); dbb = torch.zeros(512, device=dev)
    t1 = timeit(lambda: ops.call("sbl_add_layernorm_fwd", x.data_ptr(), r.data_ptr(), g.data_ptr(), be.data_ptr(), y.data_ptr(), mu.data_ptr(), rs.data_ptr(), M, 512, 1e-5, 0.1, seed.data_ptr(), 3, ops._s()))
    t2 = timeit(lambda: ops.call("sbl_add_layernorm_bwd", y.data_ptr(), x.data_ptr(), r.data_ptr(), g.data_ptr(), mu.data_ptr(), rs.data_ptr(), dz.data_ptr(), dxd.data_ptr(), dg.data_ptr(), dbb.data_ptr(), M, 512, 0.1, seed.data_ptr(), 3, ops._s()))
    arr = (ctypes.c_int * len(segL))(*segL)
    qkv = torch.randn(M, 1536, device=dev); o = torch.empty(M, 512, device=dev); do = torch.randn(M, 512, device=dev); dqkv = torch.empty_like(qkv)
    p = torch.empty(sum(H * B * L * L for L in segL), device=dev)
    t3 = timeit(lambda: ops.call("sbl_attention_seg_fwd", qkv.data_ptr(), 1536, qkv[:, 512:].data_ptr(), 1536, qkv[:, 1024:].data_ptr(), 1536, o.data_ptr(), 512, p.data_ptr(), 1, None, B, H, arr, len(segL), 0, 0.125, 0.1, seed.data_ptr(), 5, ops._s()))
    t4 = timeit(lambda: ops.call("sbl_attention_seg_bwd", do.data_ptr(), 512, qkv.data_ptr(), 1536, qkv[:, 512:].data_ptr(), 1536, qkv[:, 1024:].data_ptr(), 1536, p.data_ptr(), dqkv.data_ptr(), 1536, dqkv[:, 512:].data_ptr(), 1536, dqkv[:, 1024:].data_ptr(), 1536, B, H, arr, len(segL), 0, 0.125, 0.1, seed.data_ptr(), 5, ops._s()))
    kv = torch.randn(B * 29, 1024, device=dev); q = torch.randn(M, 512, device=dev); pc = torch.empty(H * B * sum(segL) * 29, device=dev)
    dq = torch.empty_like(q); dkv = torch.zeros_like(kv)
    t5 = timeit(lambda: ops.call("sbl_attention_seg_fwd", q.data_ptr(), 512, kv.data_ptr(), 1024, kv[:, 512:].data_ptr(), 1024, o.data_ptr(), 512, pc.data_ptr(), 0, None, B, H, arr, len(segL), 29, 0.125, 0.1, seed.data_ptr(), 5, ops._s()))
    t6 = timeit(lambda: ops.call("sbl_attention_seg_bwd", do.data_ptr(), 512, q.data_ptr(), 512, kv.data_ptr(), 1024, kv[:, 512:].data_ptr(), 1024, pc.data_ptr(), dq.data_ptr(), 512, dkv.data_ptr(), 1024, dkv[:, 512:].data_ptr(), 1024, B, H, arr, len(segL), 29, 0.125, 0.1, seed.data_ptr(), 5, ops._s()))
    print("segL=%s M=%d: LN fwd %.1f bwd %.1f | self-attn fwd %.1f bwd %.1f | cross-attn fwd %.1f bwd %.1f us" % (segL, M, t1, t2, t3, t4, t5, t6), flush=True)
