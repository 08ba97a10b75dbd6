// fp32 MFMA tile engine shared by the dense GEMMs (Linear fwd/bwd) and the implicit-GEMM
// convolutions (ResNet trunk fwd / dgrad / wgrad).
//
//   C[M,N] = sum_k A(m,k) * B(n,k)        (both operands are presented "k-major" in LDS)
//
// * 256-thread workgroup = 4 wavefronts (64 lanes) laid out 2x2; each wave owns a
//   (BM/2)x(BN/2) sub-tile made of 32x32 v_mfma_f32_32x32x2_f32 accumulators — exact fp32
//   (bitwise an fmaf chain), 157 TF peak on gfx950, which is what the 1e-3 parity bar needs.
// * LDS image As[BK][BM+4] / Bs[BK][BN+4] (k-major): the MFMA operand read
//   As[2*ks + (lane>>5)][m + (lane&31)] is 32 consecutive dwords per half-wave => conflict-free
//   ds_read_b32; two LDS buffers, one barrier per K-step, next tile's global loads issued
//   before the MFMA block and written to the other buffer after it (register staging: the
//   conv gathers are per-lane scattered, so LDS-DMA's lane-linear image does not apply).
// * Loaders are functors: dense k-contiguous rows (transposed on the LDS write), dense
//   m-contiguous rows (16-byte LDS writes), and NHWC im2col gathers of both kinds.
// * Epilogues are functors: store (+bias, +ReLU, +=), float-atomic accumulate (split-K),
//   and store + per-channel sum / sum-of-squares (training BatchNorm statistics fused into
//   the producing convolution, double atomics).
#pragma once
#include "sbl_common.h"

#include "tile_loaders.h"

// ------------------------------------------------------------------ epilogues
// MODE 0: C = acc (+bias) (ReLU)   MODE 1: C += acc (non-atomic; one block per tile)
// MODE 2: atomicAdd(C, acc) (split-K; C pre-zeroed or accumulating)
// FUSE: the residual addend and the second BatchNorm of a BasicBlock's backward (fields add_src / bs_x2 below) are compiled in
// only for the input-gradient launches that use them; the other instantiations keep their register budget.
template <int MODE, bool STATS, bool FUSE = false>
struct EpiStore {
    static constexpr bool kStats = STATS;
    static constexpr bool kFuse = FUSE;
    float* C;
    long ldc;
    const float* bias;   // [N] or nullptr
    int relu;
    double* stats;       // [2*N] (sum, sumsq) when STATS
    const float* relu_mask;   // multiply by (mask[m*ldm+n] > 0) (ReLU backward) or nullptr
    long ldm;
    // optional output row map for the parity-class input gradients: GEMM row m = (img, a, b) of the class sub-grid
    // (ca x cb) is pixel (2a + ph, 2b + pw) of the (cH x cW) image; cmap == 0: row m is output row m;
    // cmap == 2 (position-major rows): GEMM row m = pos * ca + img is output row img * cb + pos
    int cmap, ca, cb, cH, cW, cph, cpw;
    // STATS with bs_y set: instead of (sum v, sum v^2) the columns reduce the two sums of the NEXT BatchNorm backward
    // over this launch's output, g = v * (bs_y > 0) and g * (bs_x - bs_mean) * bs_inv (bs_y / bs_x laid out like C): the
    // input gradient of conv2 is the output gradient of relu(bn1(.)), so bn1's reduction pass rides on this epilogue.
    const float* bs_y;
    const float* bs_x;
    const float* bs_mean;
    const float* bs_inv;
    // a second BatchNorm fed by the same activation (the 1x1 downsample branch of a BasicBlock, video_frontend.py:69-71):
    // stats[2N..4N) = (sum g, sum g * xhat2), so both reductions of that block ride on one epilogue
    const float* bs_x2;
    const float* bs_mean2;
    const float* bs_inv2;
    // addend: v += add_src[...] before the store and the sums - the residual branch's gradient joins the input gradient
    // of conv1 here instead of in an autograd add.  add_ld == 0: laid out like C (same offset); else row m of the GEMM is
    // row m of a compact (M x N, row stride add_ld) buffer (the 1x1 / stride-2 downsample's gradient lives on the
    // even/even pixels only, i.e. on the rows of parity class (0,0)).
    const float* add_src;
    long add_ld;
    __device__ __forceinline__ long row_off(int m) const {
        if (!cmap) return (long)m * ldc;
        if (cmap == 2) {
            const int pos = m / ca;
            return ((long)(m - pos * ca) * cb + pos) * ldc;
        }
        const int img = m / (ca * cb);
        const int rem = m - img * (ca * cb);
        const int a = rem / cb, b = rem - a * cb;
        return (((long)img * cH + 2 * a + cph) * cW + 2 * b + cpw) * ldc;
    }
    __device__ __forceinline__ void tile(const f32x16& a, int mbase, int n, int M, int N, int lane,
                                         float& s1, float& s2, float& s3) const {
        if (n >= N) return;
        const float b = bias ? bias[n] : 0.f;
        float bmu = 0.f, bis = 0.f, bmu2 = 0.f, bis2 = 0.f;
        if (STATS && bs_y) {
            bmu = bs_mean[n];
            bis = bs_inv[n];
            if (FUSE && bs_x2) {
                bmu2 = bs_mean2[n];
                bis2 = bs_inv2[n];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (the fused form reads up to four more tensors per element: without a fence per row the scheduler hoists all the
            // loads of a tile - 256 VGPRs, one wave per SIMD on the 128x128 tiles)
            if (FUSE) __builtin_amdgcn_sched_barrier(0);
            const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) {
                float v = a[r] + b;
                if (relu) v = fmaxf(v, 0.f);
                if (relu_mask) v = relu_mask[(long)m * ldm + n] > 0.f ? v : 0.f;
                const long o = row_off(m) + n;
                if (FUSE && add_src) v += add_ld ? add_src[(long)m * add_ld + n] : add_src[o];
                float* q = C + o;
                if (MODE == 0) *q = v;
                else if (MODE == 1) *q += v;
                else atomicAdd(q, v);
                if (STATS) {
                    if (bs_y) {
                        const float g = bs_y[o] > 0.f ? v : 0.f;
                        s1 += g;
                        s2 += g * ((bs_x[o] - bmu) * bis);
                        if (FUSE && bs_x2) s3 += g * ((bs_x2[o] - bmu2) * bis2);
                    } else {
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------ kernel
// Split-K control.  When gridDim.z > 1 and `slabs` is set, every K-slice workgroup stores its partial tile to
// its slab, publishes it (agent-scope release) and draws a ticket from the tile's counter; the workgroup that
// draws the last ticket acquires, sums the slabs in slice order (deterministic) and runs the full epilogue
// (bias / ReLU / += ...), then re-arms the counter.  One launch, no memset, no atomics on C; placement- and
// dispatch-order independent (cdna_hip_programming.md, Guideline 16 counter form).  Counters must be zero when
// the workspace is first handed over; they are left zero by every launch.
// `a_colsum`: optional [M] vector receiving sum_k A(m,k) via float atomics (the bias gradient rides on the
// weight-gradient GEMM's A = dY^T operand); only loaders with kColSum (DenseMC) feed it.
struct SplitCtl {
    float* slabs;
    int* counters;
    float* a_colsum;
    unsigned long long* stamp;   // bench instrumentation slot or nullptr
};

// KU = BK-deep sub-tiles staged per barrier ("macro step" = KU*16 of K).  KU = 1 for the big, occupancy-rich
// problems (trunk convolutions: 3 workgroups per CU hide the load latency); KU = 4 for the decoder's skinny
// GEMMs, which run ~1 workgroup per CU and are bound by the global-load latency of each barrier-to-barrier
// step: 4x the bytes in flight per step, 4x fewer exposed latencies and barriers.
// WN = wavefronts along N (2: the 2x2 layout; 1: four wavefronts stacked along M, e.g. a 256x64 tile whose waves
// each own 64x64 = the LDS-read intensity of a 128x128 tile for the 64-channel layers).
// The body of one workgroup: output tile at (m0, n0), K range [kbeg, kend); (tile, z, nz) identify the split-K
// slice for the in-launch slab reduction (nz == 1: no split), colsum_tile: this workgroup feeds a_colsum.
// Shared tail of a tile: bias-gradient column sums, in-launch split-K slab reduction, epilogue.
template <class AL, class EPI, int BM, int BN, int WN, int TM, int TN>
__device__ __forceinline__ void sbl_tile_finish(const AL& al, const typename AL::State& sa, const EPI& epi, const SplitCtl& sc,
                                                f32x16 (&acc)[TM][TN], const float4& cs, bool do_colsum, int M, int N, int m0,
                                                int n0, int tile, int z, int nz) {
    __shared__ int s_last;
    constexpr int WM = 4 / WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    if (do_colsum) {
        const int c = al.col(sa);
        if (c + 0 < M) atomicAdd(sc.a_colsum + c + 0, cs.x);
        if (c + 1 < M) atomicAdd(sc.a_colsum + c + 1, cs.y);
        if (c + 2 < M) atomicAdd(sc.a_colsum + c + 2, cs.z);
        if (c + 3 < M) atomicAdd(sc.a_colsum + c + 3, cs.w);
    }

    // in-launch split-K reduction (last-arriving workgroup of each tile)
    if (nz > 1 && sc.slabs != nullptr) {
        float* mine = sc.slabs + ((long)tile * nz + z) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep: hipcc may drop the fence's own wait
            const int t = __hip_atomic_fetch_add(sc.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == nz - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(sc.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) {
            sbl_stamp_end(sc.stamp);
            return;
        }
        const float* base = sc.slabs + (long)tile * nz * (BM * BN) + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                for (int z = 0; z < nz; ++z) {      // slice order: deterministic sum; 16 independent loads per slice
                    const float* q = base + (long)z * (BM * BN) + (i * TN + j) * 16 * 256;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = q[r * 256];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += v[r];
                }
            }
    }

    // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) epi.tile(acc[i][j], m0 + wm * (BM / WM) + i * 32, n, M, N, lane, s1, s2, s3);
        if (EPI::kStats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32 && n < N) {
                atomicAdd(epi.stats + n, (double)s1);
                atomicAdd(epi.stats + N + n, (double)s2);
            }
            if (EPI::kFuse && epi.bs_x2) {      // (workgroup-uniform)
                s3 += __shfl_xor(s3, 32, 64);
                if (lane < 32 && n < N) {
                    atomicAdd(epi.stats + 2 * N + n, (double)s1);
                    atomicAdd(epi.stats + 3 * N + n, (double)s3);
                }
            }
        }
    }
    sbl_stamp_end(sc.stamp);
}

template <class AL, class BL, class EPI, int BM, int BN, int KU, int WN>
__device__ __forceinline__ void sbl_gemm_tile_f32(const AL& al, const BL& bl, const EPI& epi, const SplitCtl& sc, int M, int N,
                                              int m0, int n0, int kbeg, int kend, int tile, int z, int nz,
                                              bool colsum_tile) {
    constexpr int MK = KU * SBL_BK;   // macro step
    __shared__ __attribute__((aligned(16))) float As[2][MK][BM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][MK][BN + 4];
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    sbl_stamp_begin(sc.stamp);

    typename AL::State sa;
    typename BL::State sb;
    typename AL::Regs ra[KU];
    typename BL::Regs rb[KU];
    al.init(sa, m0, tid);
    bl.init(sb, n0, tid);
    const bool do_colsum = AL::kColSum && sc.a_colsum != nullptr && colsum_tile;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int u = 0; u < KU; ++u) {
        al.load(sa, kbeg + u * SBL_BK, kend, ra[u]);
        bl.load(sb, kbeg + u * SBL_BK, kend, rb[u]);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        if (do_colsum) al.accum(ra[u], cs);
        al.store(&As[0][u * SBL_BK], sa, ra[u], tid);
        bl.store(&Bs[0][u * SBL_BK], sb, rb[u], tid);
    }
    __syncthreads();

    const int arow = wm * (BM / WM) + (lane & 31);
    const int brow = wn * (BN / WN) + (lane & 31);
    const int kh = lane >> 5;
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += MK) {
        const bool has_next = (k0 + MK) < kend;
        if (has_next) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                al.load(sa, k0 + MK + u * SBL_BK, kend, ra[u]);
                bl.load(sb, k0 + MK + u * SBL_BK, kend, rb[u]);
            }
        }
        // fragment ring: the LDS reads of k-pair ks+PD-1 are issued before the MFMAs of k-pair ks, so a wave that is
        // alone on its SIMD still overlaps LDS latency with its own MFMAs (counted lgkmcnt instead of lgkmcnt(0))
        constexpr int PD = (TM * TN >= 4) ? 2 : 4;
        float a[PD][TM], b[PD][TN];
#pragma unroll
        for (int q = 0; q < PD - 1; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = As[cur][2 * q + kh][arow + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = Bs[cur][2 * q + kh][brow + j * 32];
        }
#pragma unroll
        for (int ks = 0; ks < MK / 2; ++ks) {
            if (ks + PD - 1 < MK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(ks + PD - 1) % PD][i] = As[cur][2 * (ks + PD - 1) + kh][arow + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(ks + PD - 1) % PD][j] = Bs[cur][2 * (ks + PD - 1) + kh][brow + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks % PD][i], b[ks % PD][j], acc[i][j], 0, 0, 0);
        }
        if (has_next) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                if (do_colsum) al.accum(ra[u], cs);
                al.store(&As[cur ^ 1][u * SBL_BK], sa, ra[u], tid);
                bl.store(&Bs[cur ^ 1][u * SBL_BK], sb, rb[u], tid);
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    sbl_tile_finish<AL, EPI, BM, BN, WN, TM, TN>(al, sa, epi, sc, acc, cs, do_colsum, M, N, m0, n0, tile, z, nz);
}

#include "bf16_tile.h"

// PREC = 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32); PREC = 1 / 3 / 6: bf16 MFMA on 1 / 2 / 3 bf16 planes per operand
// (bf16_tile.h: 6 = every product of an exact 3-way split whose weight is >= 2^-16, i.e. fp32-grade results).
template <class AL, class BL, class EPI, int BM, int BN, int KU, int WN, int PREC = 0, int NH = 1>
__device__ __forceinline__ void sbl_gemm_tile(const AL& al, const BL& bl, const EPI& epi, const SplitCtl& sc, int M, int N,
                                              int m0, int n0, int kbeg, int kend, int tile, int z, int nz,
                                              bool colsum_tile) {
    static_assert(PREC != 0 || NH == 1, "the fp32 body has no wave-group K split");
    if constexpr (PREC == 0)
        sbl_gemm_tile_f32<AL, BL, EPI, BM, BN, KU, WN>(al, bl, epi, sc, M, N, m0, n0, kbeg, kend, tile, z, nz, colsum_tile);
    else      // the bf16 bodies always stage one 16-deep slab per barrier: at 6 / 3 / 1 MFMAs per 32x32x16 block they are bound by
              // operand conversion and load latency, which more resident workgroups (less LDS each) hide better than longer
              // macro steps (4352x512x512 alone: 29 us at KU = 1 against 37 us at KU = 2)
        sbl_gemm_tile_bf<AL, BL, EPI, BM, BN, 1, WN, PREC, NH>(al, bl, epi, sc, M, N, m0, n0, kbeg, kend, tile, z, nz, colsum_tile);
}

// Matrix-product precision of the tile engine (sbl_set_matmul_precision): 0 = fp32 MFMA, 6 / 3 / 1 = bf16 MFMA terms.
extern int g_sbl_prec;
extern int g_sbl_wave_ksplit;      // A/B knobs (sbl_set_tuning): wave-group K split on / off,
extern int g_sbl_ksplit_tiles;     // ... the largest tile count that takes it,
extern int g_sbl_big_min;          // ... and the 64x64-tile count from which dense products take 128x128 tiles
#define SBL_PREC_LAUNCH(KERNEL_P, grid, s, ...)                                                         \
    do {                                                                                                \
        switch (g_sbl_prec) {                                                                           \
            case 6: hipLaunchKernelGGL((KERNEL_P(6)), grid, dim3(256), 0, s, __VA_ARGS__); break;       \
            case 3: hipLaunchKernelGGL((KERNEL_P(3)), grid, dim3(256), 0, s, __VA_ARGS__); break;       \
            case 1: hipLaunchKernelGGL((KERNEL_P(1)), grid, dim3(256), 0, s, __VA_ARGS__); break;       \
            default: hipLaunchKernelGGL((KERNEL_P(0)), grid, dim3(256), 0, s, __VA_ARGS__); break;      \
        }                                                                                               \
    } while (0)

// Same for kernels with a wave-group K split (last template argument NH): the split-bf16 bodies of 64x64 tiles launch 512
// threads (NH = 2) when the launch has no split-K over workgroups; the fp32 body always runs NH = 1.
#define SBL_PREC_LAUNCH_NH(KERNEL_PN, nh, grid, s, ...)                                                        \
    do {                                                                                                       \
        if ((nh) == 2 && g_sbl_prec != 0) {                                                                    \
            switch (g_sbl_prec) {                                                                              \
                case 6: hipLaunchKernelGGL((KERNEL_PN(6, 2)), grid, dim3(512), 0, s, __VA_ARGS__); break;      \
                case 3: hipLaunchKernelGGL((KERNEL_PN(3, 2)), grid, dim3(512), 0, s, __VA_ARGS__); break;      \
                default: hipLaunchKernelGGL((KERNEL_PN(1, 2)), grid, dim3(512), 0, s, __VA_ARGS__); break;     \
            }                                                                                                  \
        } else {                                                                                               \
            switch (g_sbl_prec) {                                                                              \
                case 6: hipLaunchKernelGGL((KERNEL_PN(6, 1)), grid, dim3(256), 0, s, __VA_ARGS__); break;      \
                case 3: hipLaunchKernelGGL((KERNEL_PN(3, 1)), grid, dim3(256), 0, s, __VA_ARGS__); break;      \
                case 1: hipLaunchKernelGGL((KERNEL_PN(1, 1)), grid, dim3(256), 0, s, __VA_ARGS__); break;      \
                default: hipLaunchKernelGGL((KERNEL_PN(0, 1)), grid, dim3(256), 0, s, __VA_ARGS__); break;     \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

// When the wave-group K split pays (measured inside the step, same shapes with and without it): launches of about one tile
// per CU, where a tile's serial slab chain bounds the launch - 2 x 960 x 512 x 2048 58 -> 39 us, 2 x 960 x 512 x 512
// 18 -> 15 us.  At two tiles per CU single launches still gain (4352 x 512 x 1536 110 -> 91 us) but the step does not (the
// two decoder directions' launches share the chip, and 48 KB / 512-thread workgroups co-reside worse); with 8.5 tiles per CU
// (4352 x 2048 x 512) the large workgroups halve the resident tiles: 129 -> 206 us.  Whole-step A/B on one box: threshold
// 0 / 320 / 768 tiles -> 32.99 / 32.81 / 32.99 ms.  Never with split-K over workgroups (the slab reduction has barriers).
static inline bool sbl_wave_ksplit_pays(long tiles, int nz) { return nz == 1 && g_sbl_wave_ksplit && tiles <= g_sbl_ksplit_tiles; }

// XCD-aware tile order (x_off < 0 selects it; the real offset is then -x_off - 1).  Workgroups go to the 8 XCDs
// round-robin by linear id, each XCD with its own L2.  With the plain (x fastest) order the n-tiles that share one
// A row block land on different XCDs and every L2 fetches that block again.  Here XCD c works through a contiguous
// range of the n-fastest tile order, so an A row block is fetched by one L2 and only the (small) B operand by all 8.
template <class AL, class BL, class EPI, int BM, int BN, int KU, int WN = 2, int PREC = 0, int NH = 1>
__global__ __launch_bounds__(256 * NH) void sbl_mfma_gemm_kernel(AL al, BL bl, EPI epi, SplitCtl sc, int M, int N, int K,
                                                                 int kchunk, int x_off) {
    int x = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x_off < 0) {
        // over the whole 3-D grid: the K-slices z of a weight-gradient launch share nothing, but the n-tiles of one
        // slice read the same pixels (dY chunk, x chunk shifted by a tap), so (x, y) stay together inside z
        const int X = gridDim.x, Y = gridDim.y, XY = X * Y, T = XY * gridDim.z;
        const int id = (z * Y + y) * X + x, per = T >> 3, rem = T & 7;
        const int c = id & 7, k = id >> 3;
        const int t = (c < rem ? c * (per + 1) : rem * (per + 1) + (c - rem) * per) + k;
        z = t / XY;
        const int r = t - z * XY;
        x = r / Y;
        y = r - x * Y;
        x_off = -x_off - 1;
    }
    const int kbeg = z * kchunk;
    sbl_gemm_tile<AL, BL, EPI, BM, BN, KU, WN, PREC, NH>(al, bl, epi, sc, M, N, (x + x_off) * BM, y * BN, kbeg,
                                                         min(K, kbeg + kchunk), y * gridDim.x + x, z, gridDim.z, y == 0);
}

// Two problems of one shape in one launch (the two decoder directions; see SkinnyDual): blockIdx.y in [Y, 2Y) works on
// the second operand set.  Dense loaders and the plain-store epilogue only.
struct GemmDual {
    const float* A1;
    const float* B1;
    float* C1;
    const float* bias1;
};
template <class AL, class BL, class EPI, int BM, int BN, int KU, int PREC = 0, int NH = 1>
__global__ __launch_bounds__(256 * NH) void sbl_mfma_gemm2_kernel(AL al, BL bl, EPI epi, SplitCtl sc, GemmDual du, int M, int N, int K,
                                                                  int kchunk) {
    int x = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
    {   // XCD-aware order as in sbl_mfma_gemm_kernel, over both problems
        const int X = gridDim.x, Y2 = gridDim.y, XY = X * Y2, T = XY * gridDim.z;
        const int id = (z * Y2 + y) * X + x, per = T >> 3, rem = T & 7;
        const int c = id & 7, k = id >> 3;
        const int t = (c < rem ? c * (per + 1) : rem * (per + 1) + (c - rem) * per) + k;
        z = t / XY;
        const int r = t - z * XY;
        x = r / Y2;
        y = r - x * Y2;
    }
    const int tile = y * gridDim.x + x;      // unique over both problems (split-K slabs / counters)
    const int Y = gridDim.y >> 1;
    if (y >= Y) {
        y -= Y;
        al.p = du.A1;
        bl.p = du.B1;
        epi.C = du.C1;
        epi.bias = du.bias1;
    }
    const int kbeg = z * kchunk;
    sbl_gemm_tile<AL, BL, EPI, BM, BN, KU, 2, PREC, NH>(al, bl, epi, sc, M, N, x * BM, y * BN, kbeg, min(K, kbeg + kchunk), tile, z,
                                                        gridDim.z, false);
}
template <class AL, class BL, class EPI, int BM, int BN, int KU>
static inline void sbl_launch_gemm2(const AL& al, const BL& bl, const EPI& epi, const GemmDual& du, int M, int N, int K, int splits,
                                    hipStream_t s, SplitCtl sc) {
    constexpr int MK = KU * SBL_BK;
    int kchunk = sbl_cdiv(sbl_cdiv(K, splits), MK) * MK;
    int nz = sbl_cdiv(K, kchunk);
    dim3 grid(sbl_cdiv(M, BM), 2 * sbl_cdiv(N, BN), nz);
    if constexpr (BM == 64 && BN == 64) {
#define SBL_K_(P, H) sbl_mfma_gemm2_kernel<AL, BL, EPI, BM, BN, KU, P, H>
        SBL_PREC_LAUNCH_NH(SBL_K_, sbl_wave_ksplit_pays((long)grid.x * grid.y, nz) ? 2 : 1, grid, s, al, bl, epi, sc, du, M, N, K, kchunk);
#undef SBL_K_
    } else {
#define SBL_K_(P) sbl_mfma_gemm2_kernel<AL, BL, EPI, BM, BN, KU, P>
        SBL_PREC_LAUNCH(SBL_K_, grid, s, al, bl, epi, sc, du, M, N, K, kchunk);
#undef SBL_K_
    }
}

// Position-major convolution tiles (ConvGatherPM x DenseKCTapList): the workgroup's tap list = union of the in-bounds
// taps of the (at most two, when NIMG >= BM) positions its rows cover.
template <bool DGRAD>
__device__ __forceinline__ unsigned sbl_pm_tap_mask(const ConvGeom& g, int pos) {
    const int oh = pos / g.OW, ow = pos - oh * g.OW;
    return conv_tap_mask<DGRAD, false>(g, oh, ow, g.KH * g.KW);
}
// (Measured and not kept: a persistent grid drawing tiles from an atomic counter to balance the 4 / 6 / 9-tap tiles.
// With only 1-2 tiles per resident workgroup the per-slot quantisation costs more than the balance gains: layer 3
// forward 360 -> 454 us, layer 4 390 -> 411 us.)
template <class AL, class BL, class EPI, int BM, int BN, bool DGRAD, int PREC = 0>
__global__ __launch_bounds__(256) void sbl_conv_pm_kernel(AL al, BL bl, EPI epi, SplitCtl sc, int M, int N) {
    // grid = (n-tiles, m-tiles), plain order.  (Measured and not kept: the XCD-aware order of sbl_mfma_gemm_kernel.  It
    // hands each XCD a contiguous range of positions, i.e. mostly one tap count, and the 9-tap XCD then outlasts the
    // others: layer 3 forward 362 -> 505 us although the fabric-side fetch halves.)
    const int m0 = blockIdx.y * BM;
    const int p_lo = m0 / al.g.NIMG, p_hi = (min(m0 + BM, M) - 1) / al.g.NIMG;
    unsigned mask = 0;
    for (int ps = p_lo; ps <= p_hi; ++ps) mask |= sbl_pm_tap_mask<DGRAD>(al.g, ps);
    unsigned long long list = 0;
    int nt = 0;
    for (int t = 0; t < al.g.KH * al.g.KW; ++t)
        if ((mask >> t) & 1u) {
            list |= (unsigned long long)t << (4 * nt);
            ++nt;
        }
    al.taps = list;
    bl.taps = list;
    sbl_gemm_tile<AL, BL, EPI, BM, BN, 1, 2, PREC>(al, bl, epi, sc, M, N, m0, blockIdx.x * BN, 0, nt * al.g.C,
                                                   blockIdx.y * gridDim.x + blockIdx.x, 0, 1, false);
}
// Position-major weight-gradient tiles (DenseMCPM x ConvGatherMCPM, C % BN == 0: one tap per tile); gridDim.z
// workgroups share the tile's own K' = (in-bounds pixels of its tap) * NIMG evenly and add with float atomics.
template <class AL, class BL, class EPI, int BM, int BN, int PREC = 0>
__global__ __launch_bounds__(256) void sbl_conv_pm_wgrad_kernel(AL al, BL bl, EPI epi, SplitCtl sc, int M, int N) {
    const ConvGeom& g = bl.g;
    // (plain order: the XCD-aware order measured 4 % slower here - 384 -> 400 us on layer 3)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int n0 = by * BN;
    const int tap = n0 / g.C;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    PmRect rc;
    rc.dh = kh - g.pad;
    rc.dw = kw - g.pad;
    rc.oh_lo = max(0, -rc.dh);
    rc.ow_lo = max(0, -rc.dw);
    const int nh = min(g.OH, g.SH - rc.dh) - rc.oh_lo;
    rc.nw = min(g.OW, g.SW - rc.dw) - rc.ow_lo;
    rc.rnw = 1.0f / (float)rc.nw;
    const int K = nh * rc.nw * g.NIMG;
    const int per = (K + (int)gridDim.z - 1) / (int)gridDim.z;
    const int kchunk = (per + SBL_BK - 1) / SBL_BK * SBL_BK;
    const int kbeg = bz * kchunk;
    if (kbeg >= K) return;
    al.rc = rc;
    bl.rc = rc;
    sbl_gemm_tile<AL, BL, EPI, BM, BN, 1, 2, PREC>(al, bl, epi, sc, M, N, bx * BM, n0, kbeg, min(K, kbeg + kchunk),
                                                   by * gridDim.x + bx, 0, 1, false);
}

// The four parity classes of a stride-2 input gradient (conv.hip) as ONE launch: the classes are independent dense implicit
// GEMMs with their own row grid, tap list (1 + 2 + 2 + 4 of the nine taps: K = 1, 2, 2, 4 x Cout), output row map and - class
// (0, 0) only - addend; launched one after the other the short-K classes (8-16 slabs per tile) run mostly prologue and
// epilogue while the chip drains between launches.  Here a workgroup finds its class from the tile ranges t0[] (heaviest
// class first) and runs the ordinary tile body; the per-class operand descriptors travel as kernel arguments.
#define SBL_MAX_CLASSES 4
template <class AL, class BL, class EPI>
struct ClassSet {
    AL al[SBL_MAX_CLASSES];
    BL bl[SBL_MAX_CLASSES];
    EPI epi[SBL_MAX_CLASSES];
    int M[SBL_MAX_CLASSES], K[SBL_MAX_CLASSES], t0[SBL_MAX_CLASSES + 1];
    int nclass, tiles_n;
};
template <class AL, class BL, class EPI, int BM, int BN, int PREC = 0>
__global__ __launch_bounds__(256) void sbl_conv_classes_kernel(ClassSet<AL, BL, EPI> cs, SplitCtl sc, int N) {
    const int t = blockIdx.x;
    int c = 0;
#pragma unroll
    for (int i = 1; i < SBL_MAX_CLASSES; ++i)
        if (i < cs.nclass && t >= cs.t0[i]) c = i;
    const int lt = t - cs.t0[c];
    const int mx = lt / cs.tiles_n, ny = lt - mx * cs.tiles_n;
    sbl_gemm_tile<AL, BL, EPI, BM, BN, 1, 2, PREC>(cs.al[c], cs.bl[c], cs.epi[c], sc, cs.M[c], N, mx * BM, ny * BN, 0, cs.K[c], t, 0, 1, false);
}

template <class AL, class BL, class EPI, int BM, int BN, int KU = 1, int WN = 2>
static inline void sbl_launch_gemm(const AL& al, const BL& bl, const EPI& epi, int M, int N, int K, int splits,
                                   hipStream_t s, SplitCtl sc = SplitCtl{nullptr, nullptr, nullptr, nullptr}) {
    constexpr int MK = KU * SBL_BK;
    int kchunk = sbl_cdiv(sbl_cdiv(K, splits), MK) * MK;
    int nz = sbl_cdiv(K, kchunk);
    dim3 grid(sbl_cdiv(M, BM), sbl_cdiv(N, BN), nz);
    if constexpr (BM == 64 && BN == 64 && WN == 2 && AL::kDenseOperand && BL::kDenseOperand) {
#define SBL_K_(P, H) sbl_mfma_gemm_kernel<AL, BL, EPI, BM, BN, KU, WN, P, H>
        SBL_PREC_LAUNCH_NH(SBL_K_, sbl_wave_ksplit_pays((long)grid.x * grid.y, nz) ? 2 : 1, grid, s, al, bl, epi, sc, M, N, K, kchunk, -1);
#undef SBL_K_
    } else {
#define SBL_K_(P) sbl_mfma_gemm_kernel<AL, BL, EPI, BM, BN, KU, WN, P>
        SBL_PREC_LAUNCH(SBL_K_, grid, s, al, bl, epi, sc, M, N, K, kchunk, -1);
#undef SBL_K_
    }
}

// Tail splitting.  Every tile of these launches is resident at once, so a launch lasts as long as its fullest CU:
// T tiles on 256 CUs run at ceil(T/256) tiles per CU.  Here the first floor(T/256)*256 tiles (whole M-tile columns of
// them) run unsplit and the remaining M-tile columns run as a second launch with K split `sp` ways - about one
// short workgroup per CU - reduced in-launch through the slab workspace (last arriver runs the full epilogue, so BN
// statistics / bias / ReLU work unchanged).  Returns false (nothing launched) when it does not apply.
template <class AL, class BL, class EPI, int BM, int BN, int KU = 1>
static inline bool sbl_launch_gemm_tailsplit(const AL& al, const BL& bl, const EPI& epi, int M, int N, int K, hipStream_t s,
                                             int stamp_kid, void* ws, long ws_bytes, int ws_counters) {
    constexpr int MK = KU * SBL_BK;
    const int X = sbl_cdiv(M, BM), Y = sbl_cdiv(N, BN);
    const long T = (long)X * Y;
    if (!ws || T <= 256 || T > 2048) return false;
    const int XA = (int)((T / 256) * 256 / Y);                 // whole columns of M tiles in the unsplit part
    const long rem = T - (long)XA * Y;
    if (XA <= 0 || rem <= 0 || rem > 176) return false;       // a nearly full last round gains nothing
    int sp = (int)(256 / rem);
    if (sp > K / (4 * MK)) sp = K / (4 * MK);                  // chunks of at least 4 macro steps
    if (sp > 8) sp = 8;
    if (sp < 2 || rem > ws_counters) return false;
    const long need = (long)sizeof(int) * ws_counters + rem * sp * (long)(BM * BN * sizeof(float));
    if (need > ws_bytes) return false;
    SplitCtl sa{nullptr, nullptr, nullptr, sbl_next_stamp_slot(stamp_kid)};
#define SBL_K_(P) sbl_mfma_gemm_kernel<AL, BL, EPI, BM, BN, KU, 2, P>
    SBL_PREC_LAUNCH(SBL_K_, dim3(XA, Y, 1), s, al, bl, epi, sa, M, N, K, K, -1);
    SplitCtl sb{(float*)((char*)ws + sizeof(int) * ws_counters), (int*)ws, nullptr, sbl_next_stamp_slot(stamp_kid)};
    const int kchunk = sbl_cdiv(sbl_cdiv(K, sp), MK) * MK;
    SBL_PREC_LAUNCH(SBL_K_, dim3(X - XA, Y, sbl_cdiv(K, kchunk)), s, al, bl, epi, sb, M, N, K, kchunk, -(XA + 1));
#undef SBL_K_
    return true;
}
