// Dense fp32 GEMM entry points (nn.Linear forward / input-grad / weight-grad, bias grads).
#include <stdarg.h>
#include <stdlib.h>

#include <type_traits>
#include "mfma_gemm.h"
#include "skinny_gemm.h"

// ------------------------------------------------------------------ error text (thread-local)
static thread_local char g_err[512] = "";
void sbl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sbl_last_error(void) { return g_err; }

// ------------------------------------------------------------------ bench instrumentation (process-wide, debug only)
static unsigned long long* g_stamps = nullptr;
static int g_stamp_cap = 0, g_stamp_used = 0;
static thread_local int g_last_slot = -1, g_last_kid = 0;
unsigned long long* sbl_next_stamp_slot(int kernel_id) {
    g_last_kid = kernel_id;
    g_last_slot = -1;
    if (!g_stamps || g_stamp_used >= g_stamp_cap) return nullptr;
    g_last_slot = g_stamp_used++;
    return g_stamps + 2 * (size_t)g_last_slot;
}
extern "C" int sbl_profile_begin(uint64_t* stamps_dev, int capacity) {
    SBL_REQUIRE(stamps_dev && capacity > 0, "sbl_profile_begin: bad args");
    g_stamps = (unsigned long long*)stamps_dev;
    g_stamp_cap = capacity;
    g_stamp_used = 0;
    return 0;
}
extern "C" int sbl_profile_end(void) {
    const int used = g_stamp_used;
    g_stamps = nullptr;
    g_stamp_cap = g_stamp_used = 0;
    return used;
}
extern "C" int sbl_profile_last_slot(void) { return g_last_slot; }
extern "C" int sbl_profile_used(void) { return g_stamp_used; }      // process-wide (last_slot is per calling thread)
extern "C" int sbl_profile_last_kernel(void) { return g_last_kid; }
extern "C" int sbl_abi_version(void) { return SBL_ABI_VERSION; }

// ------------------------------------------------------------------ matrix-product precision of the tile engine
int g_sbl_prec = 0;
extern "C" int sbl_set_matmul_precision(int terms) {
    SBL_REQUIRE(terms == 0 || terms == 1 || terms == 3 || terms == 6,
                "sbl_set_matmul_precision: %d (0 = fp32 MFMA, 6 / 3 / 1 = split-bf16 MFMA terms)", terms);
    g_sbl_prec = terms;
    return 0;
}
extern "C" int sbl_get_matmul_precision(void) { return g_sbl_prec; }
int g_sbl_exp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
int g_sbl_gemm2_split_target = 256, g_sbl_gemm2_split_max = 8;      // knobs 10, 11: in-launch split-K of the two-direction decoder products
int g_sbl_group_cap = 0;           // knob 6: cap on the workgroups of the grouped weight-gradient launch (0 = one per tile)
int g_sbl_wave_ksplit = 1;
int g_sbl_ksplit_tiles = 320;      // knob 2 (same-box A/B of the whole step: 0 -> 32.99, 320 -> 32.81, 768 -> 32.99 ms)
int g_sbl_big_min = 4096;          // knob 1
extern "C" int sbl_set_tuning(int knob, int value) {
    extern int g_sbl_wg_s2_small, g_sbl_wg_target, g_sbl_conv_patch;
    if (knob >= 100 && knob < 108) {      // scratch knobs for experiments (g_sbl_exp[knob - 100]); no shipped code path reads them unless DESIGN.md says so
        g_sbl_exp[knob - 100] = value;
        return 0;
    }
    if (knob == 14) {
        extern int g_sbl_stem_fwd8;
        g_sbl_stem_fwd8 = value != 0;
        return 0;
    }
    if (knob == 13) {
        extern int g_sbl_stem_ablate;
        g_sbl_stem_ablate = value;
        return 0;
    }
    if (knob == 12) {
        extern int g_sbl_stem_wgrad_tr;
        g_sbl_stem_wgrad_tr = value != 0;
        return 0;
    }
    if (knob == 10 || knob == 11) {
        SBL_REQUIRE(value >= 1, "sbl_set_tuning: value < 1");
        (knob == 10 ? g_sbl_gemm2_split_target : g_sbl_gemm2_split_max) = value;
        return 0;
    }
    if (knob == 9) {
        extern int g_sbl_conv_patch_wgrad;
        SBL_REQUIRE(value >= 0, "sbl_set_tuning: negative value");
        g_sbl_conv_patch_wgrad = value;
        return 0;
    }
    if (knob == 8) {
        extern int g_sbl_conv_patch_imgs;
        SBL_REQUIRE(value >= 0, "sbl_set_tuning: negative value");
        g_sbl_conv_patch_imgs = value;
        return 0;
    }
    if (knob == 7) {
        extern int g_sbl_pm_wg64_maxm;
        SBL_REQUIRE(value >= 0, "sbl_set_tuning: negative value");
        g_sbl_pm_wg64_maxm = value;
        return 0;
    }
    if (knob == 6) {
        SBL_REQUIRE(value >= 0, "sbl_set_tuning: negative value");
        g_sbl_group_cap = value;
        return 0;
    }
    if (knob >= 3 && knob <= 5) {
        SBL_REQUIRE(value >= 0, "sbl_set_tuning: negative value");
        (knob == 3 ? g_sbl_wg_s2_small : knob == 4 ? g_sbl_wg_target : g_sbl_conv_patch) = value;
        return 0;
    }
    SBL_REQUIRE(knob >= 0 && knob <= 2 && value >= 0, "sbl_set_tuning: unknown knob %d / value %d (0 = wave-group K split on/off, 1 = 64x64-tile count from which dense products take 128x128 tiles, 2 = largest tile count that takes the wave-group K split)", knob, value);
    if (knob == 0) g_sbl_wave_ksplit = value != 0;
    else if (knob == 1) g_sbl_big_min = value;
    else g_sbl_ksplit_tiles = value;
    return 0;
}

// ------------------------------------------------------------------ dispatch
template <class AL, class BL, int BM, int BN, int KU>
static void launch_mode(const AL& al, const BL& bl, float* C, long ldc, const float* bias, int relu,
                        const float* relu_mask, long ldm, int mode, int M, int N, int K, int splits, SplitCtl sc,
                        hipStream_t s) {
    if (mode == 0) {
        EpiStore<0, false> e{C, ldc, bias, relu, nullptr, relu_mask, ldm};
        sbl_launch_gemm<AL, BL, EpiStore<0, false>, BM, BN, KU>(al, bl, e, M, N, K, splits, s, sc);
    } else if (mode == 1) {
        EpiStore<1, false> e{C, ldc, bias, relu, nullptr, relu_mask, ldm};
        sbl_launch_gemm<AL, BL, EpiStore<1, false>, BM, BN, KU>(al, bl, e, M, N, K, splits, s, sc);
    } else {
        EpiStore<2, false> e{C, ldc, nullptr, 0, nullptr, nullptr, 0};
        sbl_launch_gemm<AL, BL, EpiStore<2, false>, BM, BN, KU>(al, bl, e, M, N, K, splits, s, sc);
    }
}

template <bool VEC, int BM, int BN, int KU>
static void launch_trans(int transA, int transB, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                         const float* bias, int relu, const float* relu_mask, long ldm, int mode, int M, int N, int K,
                         int splits, SplitCtl sc, hipStream_t s) {
    if (!transA && transB) {
        DenseKC<BM, VEC> al{A, lda, M};
        DenseKC<BN, VEC> bl{B, ldb, N};
        launch_mode<DenseKC<BM, VEC>, DenseKC<BN, VEC>, BM, BN, KU>(al, bl, C, ldc, bias, relu, relu_mask, ldm, mode, M, N, K, splits, sc, s);
    } else if (!transA && !transB) {
        DenseKC<BM, VEC> al{A, lda, M};
        DenseMC<BN, VEC> bl{B, ldb, N};
        launch_mode<DenseKC<BM, VEC>, DenseMC<BN, VEC>, BM, BN, KU>(al, bl, C, ldc, bias, relu, relu_mask, ldm, mode, M, N, K, splits, sc, s);
    } else if (transA && !transB) {
        DenseMC<BM, VEC> al{A, lda, M};
        DenseMC<BN, VEC> bl{B, ldb, N};
        launch_mode<DenseMC<BM, VEC>, DenseMC<BN, VEC>, BM, BN, KU>(al, bl, C, ldc, bias, relu, relu_mask, ldm, mode, M, N, K, splits, sc, s);
    } else {
        DenseMC<BM, VEC> al{A, lda, M};
        DenseKC<BN, VEC> bl{B, ldb, N};
        launch_mode<DenseMC<BM, VEC>, DenseKC<BN, VEC>, BM, BN, KU>(al, bl, C, ldc, bias, relu, relu_mask, ldm, mode, M, N, K, splits, sc, s);
    }
}

#define SBL_WS_COUNTERS 4096   // ints at the head of the workspace, one per output tile

extern "C" int sbl_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B,
                            long ldb, float* C, long ldc, const float* bias, int relu, const float* relu_mask, long ldm,
                            int accumulate, float* a_colsum, void* ws, long ws_bytes, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(M > 0 && N > 0 && K > 0, "sbl_gemm_f32: non-positive dims M=%d N=%d K=%d", M, N, K);
    SBL_REQUIRE(A && B && C, "sbl_gemm_f32: null operand");
    SBL_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
                "sbl_gemm_f32: leading dimension too small (lda=%ld ldb=%ld ldc=%ld)", lda, ldb, ldc);
    SBL_REQUIRE(!relu_mask || ldm >= N, "sbl_gemm_f32: ldm=%ld < N=%d", ldm, N);
    SBL_REQUIRE(!a_colsum || transA, "sbl_gemm_f32: a_colsum needs transA=1 (A stored [K][M])");
    SBL_REQUIRE(!ws || (sbl_aligned16(ws) && ws_bytes >= (long)sizeof(int) * SBL_WS_COUNTERS), "sbl_gemm_f32: workspace unaligned or < 16 KiB");
    // float4 path: every row start 16-byte aligned and, for k-contiguous operands, K % 4 == 0
    bool vec = sbl_aligned16(A) && sbl_aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    if (!transA || transB) vec = vec && (K % 4 == 0);
    if (transA) vec = vec && (M % 4 == 0);          // m-contiguous operands: whole float4s along the row axis
    if (!transB) vec = vec && (N % 4 == 0);
    SBL_REQUIRE(sbl_fits_u32((transA ? (long)K : (long)M) * lda) && sbl_fits_u32((transB ? (long)N : (long)K) * ldb),
                "sbl_gemm_f32: operand spans more than 2 GiB (buffer descriptor range)");
    const bool plain = !bias && !relu && !relu_mask;
    const long tiles64 = (long)sbl_cdiv(M, 64) * sbl_cdiv(N, 64);
    const int big_min = g_sbl_big_min;   // tuning knob (4352x2048x512: 128x128 tiles 131 us, 64x64 115 us)
    const bool big = (M >= 1024 && N >= 256 && tiles64 >= big_min);
    // split K when the output has too few 64x64 tiles to fill 256 CUs: aim at ~256 workgroups, chunks of at
    // least one 64-deep macro step, at most 8 slices (the last-arriving workgroup reads every slab)
    int splits = 1;
    constexpr int split_tiles = 192;
    constexpr int split_target = 256;
    if (!big && tiles64 < split_tiles && K >= 128) {
        splits = (int)((split_target + tiles64 - 1) / tiles64);
        if (splits > K / 64) splits = K / 64;
        if (splits > 8) splits = 8;
        if (splits < 1) splits = 1;
    }
    SplitCtl sc{nullptr, nullptr, a_colsum, nullptr};
    int mode = accumulate ? 1 : 0;
    if (splits > 1) {
        const long need = (long)sizeof(int) * SBL_WS_COUNTERS + tiles64 * splits * (long)(64 * 64 * sizeof(float));
        if (ws && tiles64 < SBL_WS_COUNTERS && need <= ws_bytes) {   // (the last counter belongs to conv.hip's persistent kernels)
            sc.counters = (int*)ws;                                     // in-launch slab reduction, full epilogue
            sc.slabs = (float*)((char*)ws + sizeof(int) * SBL_WS_COUNTERS);
        } else if (plain) {
            mode = 2;                                                   // no workspace: float atomics on C
            if (!accumulate) SBL_HIP(hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s));
        } else {
            splits = 1;
        }
    }
    // decoder-sized products: register-only skinny kernel (one workgroup per 32x32 tile, K split over its waves)
    {
        const bool a_kc = !transA, b_kc = transB != 0;
        const bool al_ok = (!a_kc || (sbl_aligned16(A) && lda % 4 == 0)) && (!b_kc || (sbl_aligned16(B) && ldb % 4 == 0));
        const bool k_ok = (!a_kc && !b_kc) || (K % 8 == 0);
        const long tiles32 = (long)sbl_cdiv(M, 32) * sbl_cdiv(N, 32);
        constexpr int max_m = 512;
        constexpr int max_t = 2048;
        // ... and d_model x d_model products up to ~1500 rows, where the 64x64 tiling would need split-K slabs to
        // fill the chip (measured 1440x512x512: fwd 18.5 vs 21.4 us, dX 16.1 vs 23.1, dW 15.8 vs 28.3)
        constexpr int sq_rows = 1536;
        const bool shape_ok = transA ? ((K <= max_m && tiles32 <= max_t) || ((long)M * N <= 512L * 512 && K <= sq_rows))
                                     : ((M <= max_m && tiles32 <= max_t) || ((long)N * K <= 512L * 512 && M <= sq_rows));
        if (shape_ok && al_ok && k_ok && !(transA && transB)) {
            SkinnyEpi e{C, ldc, bias, relu, relu_mask, ldm, accumulate, a_colsum, sbl_next_stamp_slot(SBL_KID_SKINNY)};
            if (!transA && transB) sbl_launch_skinny<true, true>(A, lda, B, ldb, e, M, N, K, s);
            else if (!transA && !transB) sbl_launch_skinny<true, false>(A, lda, B, ldb, e, M, N, K, s);
            else sbl_launch_skinny<false, false>(A, lda, B, ldb, e, M, N, K, s);
            SBL_LAUNCH_CHECK("sbl_gemm_f32(skinny)");
            return 0;
        }
    }
    sc.stamp = sbl_next_stamp_slot(big ? SBL_KID_TILED128 : SBL_KID_TILED64);
#define SBL_GO(VEC, BM, BN, KU) \
    launch_trans<VEC, BM, BN, KU>(transA, transB, A, lda, B, ldb, C, ldc, bias, relu, relu_mask, ldm, mode, M, N, K, splits, sc, s)
    if (big) {
        constexpr int big_ku = 1;
        if (vec && big_ku == 2) SBL_GO(true, 128, 128, 2);
        else if (vec) SBL_GO(true, 128, 128, 1);
        else SBL_GO(false, 128, 128, 1);
    } else {
        // KU = 4 (69 KB of LDS, 2 workgroups per CU) while every workgroup of the launch is resident at once; beyond
        // 512 workgroups KU = 2 (35 KB, 4 per CU) keeps them all resident instead of running a second, part-filled
        // round (measured 1440x2048x512: 36.6 vs 46.9 us)
        constexpr int ku_env = 0;
        const int ku = ku_env ? ku_env : (tiles64 * splits > 512 ? 2 : 4);
        if (!vec) SBL_GO(false, 64, 64, 1);
        else if (ku == 1) SBL_GO(true, 64, 64, 1);
        else if (ku == 2) SBL_GO(true, 64, 64, 2);
        else SBL_GO(true, 64, 64, 4);
    }
#undef SBL_GO
    SBL_LAUNCH_CHECK("sbl_gemm_f32");
    return 0;
}

// ------------------------------------------------------------------ two same-shape products in one launch
// C_d[M,N] = A_d[M,K] * B_d[N,K]^T (+ bias_d) (ReLU) for d = 0, 1: the nn.Linear forward of the two decoder directions
// (SBL/transformer/decoder.py:121-156 runs layer_stack_l2r[i] and layer_stack_r2l[i] back to back on same-shape
// inputs).  Same kernels, tiles and split rules as sbl_gemm_f32, sized for the doubled workgroup count.
extern "C" int sbl_gemm2_f32(int M, int N, int K, const float* A0, const float* A1, long lda, const float* B0, const float* B1,
                             long ldb, float* C0, float* C1, long ldc, const float* bias0, const float* bias1, int relu,
                             void* ws, long ws_bytes, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(M > 0 && N > 0 && K > 0, "sbl_gemm2_f32: non-positive dims M=%d N=%d K=%d", M, N, K);
    SBL_REQUIRE(A0 && A1 && B0 && B1 && C0 && C1 && (!bias0 == !bias1), "sbl_gemm2_f32: null operand / one-sided bias");
    SBL_REQUIRE(lda >= K && ldb >= K && ldc >= N, "sbl_gemm2_f32: leading dimension too small (lda=%ld ldb=%ld ldc=%ld)", lda, ldb, ldc);
    SBL_REQUIRE(!ws || (sbl_aligned16(ws) && ws_bytes >= (long)sizeof(int) * SBL_WS_COUNTERS), "sbl_gemm2_f32: workspace unaligned or < 16 KiB");
    const bool vec = sbl_aligned16(A0) && sbl_aligned16(A1) && sbl_aligned16(B0) && sbl_aligned16(B1) && lda % 4 == 0 && ldb % 4 == 0 && K % 8 == 0;
    SBL_REQUIRE(sbl_fits_u32((long)M * lda) && sbl_fits_u32((long)N * ldb), "sbl_gemm2_f32: operand spans more than 2 GiB (buffer descriptor range)");
    const long tiles64 = (long)sbl_cdiv(M, 64) * sbl_cdiv(N, 64);
    const int big_min = g_sbl_big_min;
    const bool big = (M >= 1024 && N >= 256 && 2 * tiles64 >= big_min);
    if (!vec || big) {      // shapes the decoder forward does not produce: two plain launches
        if (int e = sbl_gemm_f32(0, 1, M, N, K, A0, lda, B0, ldb, C0, ldc, bias0, relu, nullptr, 0, 0, nullptr, ws, ws_bytes, stream)) return e;
        return sbl_gemm_f32(0, 1, M, N, K, A1, lda, B1, ldb, C1, ldc, bias1, relu, nullptr, 0, 0, nullptr, ws, ws_bytes, stream);
    }
    {
        const long tiles32 = (long)sbl_cdiv(M, 32) * sbl_cdiv(N, 32);
        // (two problems = twice the workgroups: the register-only kernel stops paying at half the rows; swept inside the step)
        constexpr int max_m = 128;
        constexpr int max_t = 2048;
        constexpr int sq_rows = 768;
        if ((M <= max_m && tiles32 <= max_t) || ((long)N * K <= 512L * 512 && M <= sq_rows)) {
            SkinnyEpi e{C0, ldc, bias0, relu, nullptr, 0, 0, nullptr, sbl_next_stamp_slot(SBL_KID_SKINNY)};
            SkinnyDual du{A1, B1, C1, bias1};
            sbl_launch_skinny<true, true>(A0, lda, B0, ldb, e, M, N, K, s, &du);
            SBL_LAUNCH_CHECK("sbl_gemm2_f32(skinny)");
            return 0;
        }
    }
    int splits = 1;
    const int split_tiles = g_sbl_exp[1] > 0 ? g_sbl_exp[1] : 192;      // (experiment knob 101)
    const int split_target = g_sbl_gemm2_split_target;
    if (2 * tiles64 < split_tiles && K >= 128) {
        splits = (int)((split_target + 2 * tiles64 - 1) / (2 * tiles64));
        if (splits > K / 64) splits = K / 64;
        if (splits > g_sbl_gemm2_split_max) splits = g_sbl_gemm2_split_max;
        if (splits < 1) splits = 1;
    }
    SplitCtl sc{nullptr, nullptr, nullptr, nullptr};
    if (splits > 1) {
        const long need = (long)sizeof(int) * SBL_WS_COUNTERS + 2 * tiles64 * splits * (long)(64 * 64 * sizeof(float));
        if (ws && 2 * tiles64 < SBL_WS_COUNTERS && need <= ws_bytes) {
            sc.counters = (int*)ws;
            sc.slabs = (float*)((char*)ws + sizeof(int) * SBL_WS_COUNTERS);
        } else {
            splits = 1;
        }
    }
    sc.stamp = sbl_next_stamp_slot(SBL_KID_TILED64);
    DenseKC<64, true> al{A0, lda, M};
    DenseKC<64, true> bl{B0, ldb, N};
    EpiStore<0, false> e{C0, ldc, bias0, relu, nullptr, nullptr, 0};
    GemmDual du{A1, B1, C1, bias1};
    constexpr int ku_env = 0;
    const int ku = ku_env ? ku_env : (2 * tiles64 * splits > 512 ? 2 : 4);
    if (ku == 2) sbl_launch_gemm2<DenseKC<64, true>, DenseKC<64, true>, EpiStore<0, false>, 64, 64, 2>(al, bl, e, du, M, N, K, splits, s, sc);
    else sbl_launch_gemm2<DenseKC<64, true>, DenseKC<64, true>, EpiStore<0, false>, 64, 64, 4>(al, bl, e, du, M, N, K, splits, s, sc);
    SBL_LAUNCH_CHECK("sbl_gemm2_f32");
    return 0;
}

// ------------------------------------------------------------------ deferred weight gradient over all decoder stages
// C[M,N] += sum_s A_s^T B_s  (A_s: rows_s x M, B_s: rows_s x N, row-major), a_colsum[m] += column sums of A.
// One launch per weight per step; split-K with float atomics (C is the persistent gradient buffer).
extern "C" int sbl_wgrad_seg_f32(int nseg, const float* const* A_ptrs, long lda, const float* const* B_ptrs, long ldb,
                                 const int* seg_rows, int M, int N, float* C, long ldc, float* a_colsum,
                                 sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(nseg >= 1 && nseg <= SBL_MAX_KSEG && A_ptrs && B_ptrs && seg_rows && C, "sbl_wgrad_seg_f32: bad segment list (nseg=%d)", nseg);
    SBL_REQUIRE(M > 0 && N > 0 && lda >= M && ldb >= N && ldc >= N && lda % 4 == 0 && ldb % 4 == 0, "sbl_wgrad_seg_f32: bad dims M=%d N=%d lda=%ld ldb=%ld", M, N, lda, ldb);
    constexpr int seg_tile_env = 0;
    constexpr int seg_ku = 2;
    constexpr int seg_target = 768;
    long K = 0;
    bool aligned = true;
    for (int t = 0; t < nseg; ++t) {
        aligned = aligned && (seg_rows[t] % SBL_BK == 0);
        SBL_REQUIRE(A_ptrs[t] && B_ptrs[t] && seg_rows[t] > 0 && sbl_aligned16(A_ptrs[t]) && sbl_aligned16(B_ptrs[t]), "sbl_wgrad_seg_f32: segment %d null/unaligned/empty", t);
        K += seg_rows[t];
    }
    SBL_REQUIRE(K < (1L << 30), "sbl_wgrad_seg_f32: too many rows");
    unsigned long long* stamp = sbl_next_stamp_slot(SBL_KID_SEG_WGRAD);
    auto go = [&](auto al, auto tile_c, auto ku_c) {
        constexpr int T = decltype(tile_c)::value, KUc = decltype(ku_c)::value;
        decltype(al) bl;
        long kc = 0;
        for (int t = 0; t < SBL_MAX_KSEG; ++t) {
            al.kcum[t] = bl.kcum[t] = (int)kc;
            al.p[t] = t < nseg ? A_ptrs[t] : nullptr;
            bl.p[t] = t < nseg ? B_ptrs[t] : nullptr;
            if (t < nseg) kc += seg_rows[t];
        }
        al.kcum[SBL_MAX_KSEG] = bl.kcum[SBL_MAX_KSEG] = (int)K;
        al.nseg = bl.nseg = nseg;
        al.ld = lda; bl.ld = ldb;
        al.rows = M; bl.rows = N;
        const long tiles = (long)sbl_cdiv(M, T) * sbl_cdiv(N, T);
        int splits = (int)((seg_target + tiles - 1) / tiles);          // ~3 workgroups per CU; chunks >= 128 rows
        if (splits > K / 128) splits = (int)(K / 128);
        if (splits < 1) splits = 1;
        EpiStore<2, false> e{C, ldc, nullptr, 0, nullptr, nullptr, 0};
        SplitCtl sc{nullptr, nullptr, a_colsum, stamp};
        sbl_launch_gemm<decltype(al), decltype(al), EpiStore<2, false>, T, T, KUc>(al, bl, e, M, N, (int)K, splits, s, sc);
    };
    using std::integral_constant;
    // 128x128 tiles (twice the flops per staged byte) once the weight has enough of them to split K over; measured at
    // K = 4352 rows: 2048x512 67 vs 52 TF, 1536x512 58 vs 50, 512x512 26 vs 34
    const int seg_tile = seg_tile_env ? seg_tile_env : ((long)sbl_cdiv(M, 128) * sbl_cdiv(N, 128) >= 48 ? 128 : 64);
    if (!aligned) go(SegMC<64, false>{}, integral_constant<int, 64>{}, integral_constant<int, 2>{});
    else if (seg_tile == 128) go(SegMC<128, true>{}, integral_constant<int, 128>{}, integral_constant<int, 1>{});
    else if (seg_ku == 4) go(SegMC<64, true>{}, integral_constant<int, 64>{}, integral_constant<int, 4>{});
    else go(SegMC<64, true>{}, integral_constant<int, 64>{}, integral_constant<int, 2>{});
    SBL_LAUNCH_CHECK("sbl_wgrad_seg_f32");
    return 0;
}

// ------------------------------------------------------------------ all deferred weight gradients in ONE launch
// The per-weight launches above have 16-64 output tiles each, so they split K 12-24 ways to fill the chip: short K
// loops, and every output element is hit by that many float atomics.  All weights of a step share the same stage row
// structure, so they can be one grouped launch: 2688 128x128 tiles for the 72 decoder weights, each tile owned by one
// workgroup over the whole K = 4352 (a long, efficient K loop), C += acc without atomics (deterministic), ~10 tiles
// per CU for balance.  Problem descriptors live in a device table written by a tiny kernel (they do not fit in 4 KB
// of kernel arguments, and a host->device copy cannot be captured into a hipGraph from pageable memory).
struct GroupProb {
    const float* a[SBL_MAX_KSEG];
    const float* b[SBL_MAX_KSEG];
    float* C;
    float* colsum;
    long lda, ldb, ldc;
    int M, N, tile0, tiles_m;
};
#define SBL_GROUP_WRITE 8
struct GroupWrite {
    GroupProb p[SBL_GROUP_WRITE];
};
__global__ void group_write_kernel(GroupWrite w, GroupProb* table, int first, int count) {
    const int i = threadIdx.x;
    if (i < count) table[first + i] = w.p[i];
}
struct GroupCommon {
    int nprob, nseg, K, ntiles;
    int kcum[SBL_MAX_KSEG + 1];
};
// ONESEG: every problem's K rows are one contiguous block (the stage-batched decoder backward, the encoder): plain
// m-contiguous loaders.  The segmented loaders, built in registers from the table, index their pointer arrays
// dynamically, which puts them in scratch memory (440 bytes per lane) and a scratch read on every operand load.
template <bool ONESEG, int PREC>
__global__ __launch_bounds__(256) void sbl_wgrad_group_kernel(const GroupProb* __restrict__ table, GroupCommon gc,
                                                              unsigned long long* stamp) {
    // (gridDim.x may be capped below the tile count - sbl_set_tuning knob 6: the workgroups then walk the tiles, so that the
    // launch takes a bounded share of the CUs while a dependent chain of small kernels runs on the other stream)
  for (int t = blockIdx.x; t < gc.ntiles; t += gridDim.x) {
    if (t != (int)blockIdx.x) __syncthreads();      // the previous tile's LDS reads are done
    // problem of this tile: binary search over tile0 (ascending), workgroup-uniform
    int lo = 0, hi = gc.nprob - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    const GroupProb& g = table[lo];
    const int lt = t - g.tile0;
    const int tx = lt % g.tiles_m, ty = lt / g.tiles_m;
    EpiStore<1, false> e{g.C, g.ldc, nullptr, 0, nullptr, nullptr, 0};
    SplitCtl sc{nullptr, nullptr, g.colsum, stamp};
    if (ONESEG) {
        DenseMC<128, true> al{g.a[0], g.lda, g.M}, bl{g.b[0], g.ldb, g.N};
        sbl_gemm_tile<DenseMC<128, true>, DenseMC<128, true>, EpiStore<1, false>, 128, 128, 1, 2, PREC>(al, bl, e, sc, g.M, g.N, tx * 128,
                                                                                                    ty * 128, 0, gc.K, 0, 0, 1, ty == 0);
    } else {
        SegMC<128, true> al, bl;
#pragma unroll
        for (int s = 0; s < SBL_MAX_KSEG; ++s) {
            al.p[s] = g.a[s];
            bl.p[s] = g.b[s];
            al.kcum[s] = bl.kcum[s] = gc.kcum[s];
        }
        al.kcum[SBL_MAX_KSEG] = bl.kcum[SBL_MAX_KSEG] = gc.K;
        al.nseg = bl.nseg = gc.nseg;
        al.ld = g.lda; bl.ld = g.ldb;
        al.rows = g.M; bl.rows = g.N;
        sbl_gemm_tile<SegMC<128, true>, SegMC<128, true>, EpiStore<1, false>, 128, 128, 1, 2, PREC>(al, bl, e, sc, g.M, g.N, tx * 128,
                                                                                                  ty * 128, 0, gc.K, 0, 0, 1, ty == 0);
    }
  }
}

extern "C" long sbl_wgrad_group_table_bytes(int nprob) { return (long)sizeof(GroupProb) * (nprob > 0 ? nprob : 0); }

extern "C" int sbl_wgrad_group_f32(int nprob, int nseg, const int* seg_rows, const float* const* A_ptrs, const long* lda,
                                   const float* const* B_ptrs, const long* ldb, const int* M, const int* N, float* const* C,
                                   const long* ldc, float* const* colsum, void* table, long table_bytes,
                                   sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(nprob >= 1 && nprob <= 4096 && nseg >= 1 && nseg <= SBL_MAX_KSEG && seg_rows && A_ptrs && B_ptrs && lda && ldb && M && N && C && ldc && colsum,
                "sbl_wgrad_group_f32: bad arguments (nprob=%d nseg=%d)", nprob, nseg);
    SBL_REQUIRE(table && sbl_aligned16(table) && table_bytes >= sbl_wgrad_group_table_bytes(nprob), "sbl_wgrad_group_f32: descriptor table too small");
    GroupCommon gc;
    gc.nprob = nprob;
    gc.nseg = nseg;
    long K = 0;
    for (int t = 0; t < SBL_MAX_KSEG; ++t) {
        gc.kcum[t] = (int)K;
        if (t < nseg) {
            SBL_REQUIRE(seg_rows[t] > 0 && seg_rows[t] % SBL_BK == 0, "sbl_wgrad_group_f32: segment %d has %d rows (must be a positive multiple of %d)", t, seg_rows[t], SBL_BK);
            K += seg_rows[t];
        }
    }
    SBL_REQUIRE(K < (1L << 30), "sbl_wgrad_group_f32: too many rows");
    gc.kcum[SBL_MAX_KSEG] = (int)K;
    gc.K = (int)K;
    long tiles = 0;
    GroupProb* tab = (GroupProb*)table;
    for (int first = 0; first < nprob; first += SBL_GROUP_WRITE) {
        GroupWrite w;
        const int count = nprob - first < SBL_GROUP_WRITE ? nprob - first : SBL_GROUP_WRITE;
        for (int i = 0; i < count; ++i) {
            const int p = first + i;
            GroupProb& g = w.p[i];
            SBL_REQUIRE(M[p] % 4 == 0 && N[p] % 4 == 0 && sbl_fits_u32(K * lda[p]) && sbl_fits_u32(K * ldb[p]),
                        "sbl_wgrad_group_f32: problem %d: M=%d / N=%d must be multiples of 4 and the operands at most 2 GiB", p, M[p], N[p]);
            SBL_REQUIRE(M[p] > 0 && N[p] > 0 && C[p] && lda[p] >= M[p] && ldb[p] >= N[p] && ldc[p] >= N[p] && lda[p] % 4 == 0 && ldb[p] % 4 == 0,
                        "sbl_wgrad_group_f32: problem %d has bad dims M=%d N=%d lda=%ld ldb=%ld ldc=%ld", p, M[p], N[p], lda[p], ldb[p], ldc[p]);
            for (int t = 0; t < SBL_MAX_KSEG; ++t) {
                g.a[t] = t < nseg ? A_ptrs[(long)p * nseg + t] : nullptr;
                g.b[t] = t < nseg ? B_ptrs[(long)p * nseg + t] : nullptr;
                SBL_REQUIRE(t >= nseg || (g.a[t] && g.b[t] && sbl_aligned16(g.a[t]) && sbl_aligned16(g.b[t])), "sbl_wgrad_group_f32: problem %d segment %d null/unaligned", p, t);
            }
            g.C = C[p];
            g.colsum = colsum[p];
            g.lda = lda[p]; g.ldb = ldb[p]; g.ldc = ldc[p];
            g.M = M[p]; g.N = N[p];
            g.tile0 = (int)tiles;
            g.tiles_m = sbl_cdiv(M[p], 128);
            tiles += (long)g.tiles_m * sbl_cdiv(N[p], 128);
        }
        hipLaunchKernelGGL(group_write_kernel, dim3(1), dim3(64), 0, s, w, tab, first, count);
    }
    SBL_REQUIRE(tiles < (1L << 30), "sbl_wgrad_group_f32: too many tiles");
    gc.ntiles = (int)tiles;
    if (g_sbl_group_cap > 0 && tiles > g_sbl_group_cap) tiles = g_sbl_group_cap;      // workgroups of the launch (they walk the tiles)
#define SBL_KG1_(P) sbl_wgrad_group_kernel<true, P>
#define SBL_KG0_(P) sbl_wgrad_group_kernel<false, P>
    if (nseg == 1)
        SBL_PREC_LAUNCH(SBL_KG1_, dim3((unsigned)tiles), s, (const GroupProb*)tab, gc, sbl_next_stamp_slot(SBL_KID_SEG_WGRAD));
    else
        SBL_PREC_LAUNCH(SBL_KG0_, dim3((unsigned)tiles), s, (const GroupProb*)tab, gc, sbl_next_stamp_slot(SBL_KID_SEG_WGRAD));
#undef SBL_KG1_
#undef SBL_KG0_
    SBL_LAUNCH_CHECK("sbl_wgrad_group_f32");
    return 0;
}

// ------------------------------------------------------------------ column sums (bias grads)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ldx, float* __restrict__ out,
                                                     int M, int N, int rows_per_block) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int r = r0 + rg; r < r1; r += 4) s += X[(long)r * ldx + c];
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int sbl_colsum_f32(const float* X, long ldx, float* out, int M, int N, int accumulate, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(M > 0 && N > 0 && X && out && ldx >= N, "sbl_colsum_f32: bad args M=%d N=%d ldx=%ld", M, N, ldx);
    if (!accumulate) SBL_HIP(hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s));
    int gy = sbl_cdiv(M, 64);
    if (gy > 64) gy = 64;
    int rpb = sbl_cdiv(M, gy);
    hipLaunchKernelGGL(colsum_kernel, dim3(sbl_cdiv(N, 64), gy), dim3(256), 0, s, X, ldx, out, M, N, rpb);
    SBL_LAUNCH_CHECK("sbl_colsum_f32");
    return 0;
}
