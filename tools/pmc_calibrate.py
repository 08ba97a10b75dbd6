"""Kernels of KNOWN HBM traffic for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM: the
counters tally fabric-side requests; wide streaming reads are reported at one half, other access shapes must be calibrated).
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and again with WRITE_SIZE); tools/pmc_report.py divides the known
bytes printed here by what the counter reports for the same kernel and applies the factor per access pattern:
  stream16   a float4-per-lane streaming read (torch copy kernel; bn_apply_fwd_kernel reads two such streams)
  tile_kc    the tile engine's k-contiguous operand loader (buffer_load_dwordx4, 4 lanes per 64-byte row segment):
             a 16384 x 64 x 4096 product reads A (268 MB) exactly once; B (1 MB) stays in L2
  tile_mc    the m-contiguous loader: a 64 x 16384 x 4096 weight-gradient-shaped product (A^T: 4096 x 64, B: 4096 x 16384 = 268 MB)
Every buffer is larger than the 256 MB Infinity Cache or touched once, so nothing is served on-die."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
known = {}
n = 1 << 28                                   # 1 GiB of fp32
x = torch.randn(n, device=dev); y = torch.empty_like(x)
flush = torch.empty(1 << 28, device=dev)
for _ in range(3):
    flush.zero_()                             # evict
    y.copy_(x)
known["stream16_copy"] = {"kernel_re": "elementwise_kernel_manual_unroll|vectorized_elementwise_kernel<4, at::native::.*copy|direct_copy", "read": 4 * n, "write": 4 * n}
rows, C = 928 * 44 * 44, 64                   # the stem's conv-out size: 460 MB
conv = torch.randn(rows, C, device=dev); out = torch.empty_like(conv)
mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
for _ in range(3):
    flush.zero_()
    ops.call("sbl_bn_apply_fwd", conv.data_ptr(), None, mean.data_ptr(), inv.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), rows, C, 1, ops._s())
known["stream16_bn_apply"] = {"kernel_re": "bn_apply_fwd_kernel", "read": 4 * rows * C, "write": 4 * rows * C}
ops.set_matmul_precision("f32")
M, N, K = 16384, 64, 4096
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); Cm = torch.empty(M, N, device=dev)
for _ in range(3):
    flush.zero_()
    ops.gemm(0, 1, M, N, K, A, K, B, K, Cm, N)
known["tile_kc"] = {"kernel_re": r"sbl_mfma_gemm_kernel<DenseKC<\d+, true>, DenseKC<\d+, true>", "read": 4 * (M * K + N * K), "write": 4 * M * N}
At = torch.randn(K, 64, device=dev); Bt = torch.randn(K, 16384, device=dev); Cw = torch.zeros(64, 16384, device=dev)
for _ in range(3):
    flush.zero_()
    ops.gemm(1, 0, 64, 16384, K, At, 64, Bt, 16384, Cw, 16384)
known["tile_mc"] = {"kernel_re": r"sbl_mfma_gemm_kernel<DenseMC<\d+, true>, DenseMC<\d+, true>", "read": 4 * (K * 64 + K * 16384), "write": 4 * 64 * 16384}
torch.cuda.synchronize()
json.dump(known, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_known.json", "w"), indent=1)
print("known traffic written")
