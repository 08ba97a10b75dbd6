// Latency-optimised fp32 MFMA GEMM for the SBL decoder's skinny products (M = 32*L <= 512 rows, prefix L <= 16).
//
// The tiled engine (mfma_gemm.h) needs one barrier and one global-load round trip per K-step, and at one
// workgroup per CU those round trips are exposed: ~15-25 us for products that hold 2-3 us of MFMA work.
// Here one workgroup owns ONE 32x32 output tile and its NW wavefronts split K; every wave feeds
// v_mfma_f32_32x32x2_f32 straight from global memory — no LDS staging, no barrier in the main loop, all loads of
// a 32-deep K chunk in flight before the first MFMA that needs them (register double buffering) — and the NW
// partial tiles meet once in LDS, where the epilogue (bias / ReLU / ReLU-mask / +=) runs.
//
// MFMA operand map (32x32x2): lane l supplies A[i = l&31][k] and B[k][j = l&31] for ONE k per instruction;
// lanes 0-31 and 32-63 supply two different k.  A lane loads 4 consecutive k at a time (float4 for k-contiguous
// operands), so MFMA s of an 8-deep group contracts k = kb + 4*(l>>5) + s — any pairing of distinct k works as
// long as A and B use the same one.
#pragma once
#include "sbl_common.h"

// operand element (r, k):  KC: p[r*ld + k]   MC: p[k*ld + r]
template <bool KCONTIG>
struct SkinnyOperand {
    const float* p;
    long ld;
    int rows;
    // 4 values for k = kb + s, s = 0..3 (kb already includes the lane's 4*(l>>5)); zero beyond kend
    __device__ __forceinline__ float4 load(int r, int kb, int kend) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONTIG) {
            if (kb < kend) v = *reinterpret_cast<const float4*>(p + (long)r * ld + kb);   // K % 8 == 0 on this path
        } else {
            const float* q = p + (long)kb * ld + r;
            if (kb + 0 < kend) v.x = q[0];
            if (kb + 1 < kend) v.y = q[ld];
            if (kb + 2 < kend) v.z = q[2 * ld];
            if (kb + 3 < kend) v.w = q[3 * ld];
        }
        return v;
    }
};

struct SkinnyEpi {
    float* C;
    long ldc;
    const float* bias;
    int relu;
    const float* relu_mask;
    long ldm;
    int accumulate;
    float* a_colsum;   // [M] += sum_k A(m,k) (float atomics) or nullptr
    unsigned long long* stamp;   // bench instrumentation slot or nullptr
};

// Second problem of the same shape (the other decoder direction: its own operands, output and bias), handled by the
// workgroups with blockIdx.z == 1 of the same launch; all nullptr / gridDim.z == 1 otherwise.  Kernel boundaries
// cost ~5 us each and launches on two streams do not overlap at these sizes (tools/bench_twochain.py), so the two
// directions of the decoder forward share launches instead of streams.
struct SkinnyDual {
    const float* A1;
    const float* B1;
    float* C1;
    const float* bias1;
};

template <bool AKC, bool BKC, int NW, int U>
__global__ __launch_bounds__(NW * 64) void sbl_skinny_gemm_kernel(SkinnyOperand<AKC> a, SkinnyOperand<BKC> b, SkinnyEpi e,
                                                                  SkinnyDual du, int M, int N, int K) {
    __shared__ float red[NW][1024];
    __shared__ float csum[NW][32];
    if (blockIdx.z) {
        a.p = du.A1;
        b.p = du.B1;
        e.C = du.C1;
        e.bias = du.bias1;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i31 = lane & 31, h = lane >> 5;
    // XCD-aware tile order (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, each
    // with its own L2.  With the natural order every XCD ends up fetching all of A and all of B from the fabric
    // (rocprofv3 FETCH_SIZE: 9-10 MB per launch against ~2 MB of operands).  Give XCD x the column tiles
    // n = x (mod 8): the B (weight) slice of a column tile is then fetched by one L2 only.
    int mt = blockIdx.x, nt = blockIdx.y;
    {
        const int MT = gridDim.x, NT = gridDim.y;
        if ((NT & 7) == 0) {
            const int b = blockIdx.y * MT + blockIdx.x;      // linear dispatch index
            const int xcd = b & 7, q = b >> 3;
            mt = q % MT;
            nt = (q / MT) * 8 + xcd;
        }
    }
    const int m0 = mt * 32, n0 = nt * 32;
    const int ra = min(m0 + i31, M - 1);     // clamped: rows beyond the edge are computed but never stored
    const int rb = min(n0 + i31, N - 1);
    // this wave's K range: contiguous, multiple of 8
    const int per = ((K + NW - 1) / NW + 7) & ~7;
    const int kbeg = wave * per;
    const int kend = min(K, kbeg + per);
    sbl_stamp_begin(e.stamp);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    const bool do_colsum = e.a_colsum != nullptr && nt == 0;

    // U = 8-deep groups per register stage: 4 (32 of K) normally, 8 for K >= 1024 so that half of a wave's K
    // range is in flight at any time (the K = 2048 FFN products were load-latency bound at U = 4)
    // two named register stages (static indexing only: runtime-indexed register arrays would go to scratch)
    float4 a0[U], b0[U], a1[U], b1[U];
#define SK_LOAD(AV, BV, KB)                                   \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {           \
        AV[u] = a.load(ra, (KB) + u * 8 + 4 * h, kend);       \
        BV[u] = b.load(rb, (KB) + u * 8 + 4 * h, kend);       \
    }
#define SK_MMA(AV, BV)                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                 \
        const float4 x = AV[u], y = BV[u];                                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc, 0, 0, 0);         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc, 0, 0, 0);         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc, 0, 0, 0);         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc, 0, 0, 0);         \
        if (do_colsum) asum += (x.x + x.y) + (x.z + x.w);                           \
    }
    SK_LOAD(a0, b0, kbeg)
    for (int k0 = kbeg; k0 < kend; k0 += 16 * U) {
        const bool has1 = k0 + 8 * U < kend;
        if (has1) { SK_LOAD(a1, b1, k0 + 8 * U) }
        SK_MMA(a0, b0)
        if (k0 + 16 * U < kend) { SK_LOAD(a0, b0, k0 + 16 * U) }
        if (has1) { SK_MMA(a1, b1) }
    }
#undef SK_LOAD
#undef SK_MMA
    // meet in LDS: red[w][r*64 + lane] (conflict-free), then every thread finishes 1024/(NW*64) outputs
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
    if (do_colsum) {
        asum += __shfl_xor(asum, 32, 64);
        if (h == 0) csum[wave][i31] = asum;
    }
    __syncthreads();
    if (do_colsum && tid < 32 && m0 + tid < M) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += csum[w][tid];
        atomicAdd(e.a_colsum + m0 + tid, s);
    }
    for (int el = tid; el < 1024; el += NW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][el];
        const int r = el >> 6, ln = el & 63;
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int n = n0 + (ln & 31);
        if (m < M && n < N) {
            if (e.bias) v += e.bias[n];
            if (e.relu) v = fmaxf(v, 0.f);
            if (e.relu_mask) v = e.relu_mask[(long)m * e.ldm + n] > 0.f ? v : 0.f;
            float* q = e.C + (long)m * e.ldc + n;
            if (e.accumulate) v += *q;
            *q = v;
        }
    }
    sbl_stamp_end(e.stamp);
}

template <bool AKC, bool BKC>
static inline void sbl_launch_skinny(const float* A, long lda, const float* B, long ldb, const SkinnyEpi& e, int M, int N,
                                     int K, hipStream_t s, const SkinnyDual* dual = nullptr) {
    SkinnyOperand<AKC> a{A, lda, M};
    SkinnyOperand<BKC> b{B, ldb, N};
    const SkinnyDual du = dual ? *dual : SkinnyDual{nullptr, nullptr, nullptr, nullptr};
    dim3 grid(sbl_cdiv(M, 32), sbl_cdiv(N, 32), dual ? 2 : 1);
    if (K >= 1024) hipLaunchKernelGGL((sbl_skinny_gemm_kernel<AKC, BKC, 8, 8>), grid, dim3(512), 0, s, a, b, e, du, M, N, K);
    else if (K >= 512) hipLaunchKernelGGL((sbl_skinny_gemm_kernel<AKC, BKC, 8, 4>), grid, dim3(512), 0, s, a, b, e, du, M, N, K);
    else hipLaunchKernelGGL((sbl_skinny_gemm_kernel<AKC, BKC, 4, 4>), grid, dim3(256), 0, s, a, b, e, du, M, N, K);
}
