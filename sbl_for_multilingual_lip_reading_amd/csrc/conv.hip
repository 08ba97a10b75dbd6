// ResNet-18 trunk convolutions as NHWC implicit GEMM on the fp32 MFMA tile engine
// (SBL/transformer/video_frontend.py:10-12 conv3x3, :69-70 1x1/stride-2 shortcut).
//   fwd   : y[pix, co]            = sum_{tap,ci} x[gather(pix,tap), ci] * w[co, tap, ci]      (+ BN stats)
//   dgrad : dx[pix, ci]           = sum_{tap,co} dy[gather'(pix,tap), co] * wt[ci, tap, co]
//   wgrad : dw[co, (tap,ci)]      = sum_{pix}    dy[pix, co] * x[gather(pix,tap), ci]         (split-K atomics)
#include "mfma_gemm.h"
#include "conv_patch.h"
#include "conv_patch_wgrad.h"

int g_sbl_pm_wg64_maxm = 512;    // knob 7: position-major weight gradients with Cout <= this on 64x64 tiles (0 = always 128x128; same-box step
                                 // A/B 0 / 128 / 256 / 512: 32.33 / 32.32 / 32.33 / 32.20 ms)
int g_sbl_conv_patch = 2;        // sbl_set_tuning knob 5: patch-resident 3x3 / stride-1 kernel for the large maps (conv_patch.h): 0 off,
                                 // 1 padded 64-channel rows (one workgroup per CU), 2 swizzled 32-channel rows (two per CU; default)
int g_sbl_conv_patch_wgrad = 30;  // knob 9: patch-resident weight gradient (conv_patch_wgrad.h) for 3x3 / stride-1 maps of at least this many pixels (0 = off;
                                 // same-box step A/B 0 / 100 / 30: 32.22 / 31.84 / 31.73 ms)
int g_sbl_conv_patch_imgs = 0;   // knob 8: most images per tile of that kernel (0 = as many as fit: two 11x11 maps; 1 = single-image tiles only, i.e. layer 1 only)
int g_sbl_wg_s2_small = 1;      // sbl_set_tuning knob 3: stride-2 weight gradients on 64x64 tiles (128 -> 256: 459 -> 335 us, 256 -> 512: 447 -> 400 us)
int g_sbl_wg_target = 1536;     // knob 4: their workgroup target (0 = the default rule; same-box step A/B 32.44 / 32.34 / 32.29 ms for 128-tiles / 64-tiles / 64-tiles + 1536)
static int check_conv(const char* who, int NIMG, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    SBL_REQUIRE(NIMG > 0 && H > 0 && W > 0, "%s: bad image dims %d %d %d", who, NIMG, H, W);
    SBL_REQUIRE(Cin % 16 == 0 && Cout % 16 == 0 && Cin >= 16 && Cout >= 16, "%s: Cin=%d Cout=%d must be multiples of 16", who, Cin, Cout);
    SBL_REQUIRE((KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0), "%s: only 3x3/pad1 and 1x1/pad0 (got %dx%d pad %d)", who, KH, KW, pad);
    SBL_REQUIRE(stride == 1 || stride == 2, "%s: stride %d", who, stride);
    SBL_REQUIRE(sbl_fits_u32((long)NIMG * H * W * (long)(Cin > Cout ? Cin : Cout)), "%s: tensor spans more than 2 GiB (buffer descriptor range)", who);
    SBL_REQUIRE((long)NIMG * H * W < (1L << 24) && Cin < (1 << 16) && Cout < (1 << 16), "%s: more than 2^24 pixels (24-bit index arithmetic in the gathers)", who);
    return 0;
}
static inline int out_dim(int H, int K, int stride, int pad) { return (H + 2 * pad - K) / stride + 1; }

// Position-major path for 3x3 / stride-1 convolutions on small maps (ResNet layers 2-4: 11x11, 6x6, 3x3): SBL_CONV_PM_HW =
// largest Ho*Wo that takes it (0 = off; A/B knob; the 22x22 maps of layer 1 lose: 6 % padding, 434 vs 396 us)
constexpr int g_pm_hw = 121;
// forward / input-gradient tile of that path: 1 = 128x128, 2 = 128x64, 3 = 64x64, 0 = the ordinary launches' rule.  The
// tiles are uneven (4 / 6 / 9 taps) and co-resident, so small tiles balance best: layer 4 forward 400 -> 307 us
// (128 TF of algorithmic FLOPs), input gradient 431 -> 324 us; layer 2 352 -> 327 us
constexpr int g_pm_tile = 3;
static inline bool conv_pm_ok(int Ho, int Wo, int KH, int stride) { return KH == 3 && stride == 1 && Ho * Wo <= g_pm_hw; }

#define SBL_CONV_WS_COUNTERS 4096      // same workspace convention as sbl_gemm_f32: int counters, then fp32 slabs
constexpr int g_tailsplit = 1;
extern "C" int sbl_conv2d_fwd(const float* x, const float* w, float* y, double* stats, int stats_zeroed, int NIMG, int H, int W,
                              int Cin, int Cout, int KH, int KW, int stride, int pad, void* ws, long ws_bytes, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_conv("sbl_conv2d_fwd", NIMG, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    SBL_REQUIRE(x && w && y && sbl_aligned16(x) && sbl_aligned16(w), "sbl_conv2d_fwd: null/unaligned pointer");
    const int Ho = out_dim(H, KH, stride, pad), Wo = out_dim(W, KW, stride, pad);
    const int M = NIMG * Ho * Wo, N = Cout, K = KH * KW * Cin;
    ConvGeom g{NIMG, Ho, Wo, H, W, Cin, KH, KW, stride, pad, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
    sbl_geom_finish(g);
    if (stats && !stats_zeroed) SBL_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * Cout, s));
    const long t128 = (long)sbl_cdiv(M, 128) * sbl_cdiv(N, 128);
    SBL_REQUIRE(!ws || (sbl_aligned16(ws) && ws_bytes >= (long)sizeof(int) * SBL_CONV_WS_COUNTERS), "sbl_conv2d_fwd: workspace unaligned or < 16 KiB");
#define SBL_CONV_FWD(BM, BN, WN)                                                                               \
    do {                                                                                                       \
        ConvGatherKC<BM, false> al{x, g, M};                                                                   \
        DenseKC<BN, true> bl{w, (long)K, N};                                                                   \
        if (stats) {                                                                                           \
            EpiStore<0, true> e{y, (long)N, nullptr, 0, stats, nullptr, 0};                                    \
            if (!(g_tailsplit && sbl_launch_gemm_tailsplit<ConvGatherKC<BM, false>, DenseKC<BN, true>, EpiStore<0, true>, BM, BN, 1>(al, bl, e, M, N, K, s, SBL_KID_CONV_FWD, ws, ws_bytes, SBL_CONV_WS_COUNTERS))) { \
                SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_FWD)};                 \
                sbl_launch_gemm<ConvGatherKC<BM, false>, DenseKC<BN, true>, EpiStore<0, true>, BM, BN, 1, WN>(al, bl, e, M, N, K, 1, s, sc); \
            }                                                                                                  \
        } else {                                                                                               \
            EpiStore<0, false> e{y, (long)N, nullptr, 0, nullptr, nullptr, 0};                                 \
            if (!(g_tailsplit && sbl_launch_gemm_tailsplit<ConvGatherKC<BM, false>, DenseKC<BN, true>, EpiStore<0, false>, BM, BN, 1>(al, bl, e, M, N, K, s, SBL_KID_CONV_FWD, ws, ws_bytes, SBL_CONV_WS_COUNTERS))) { \
                SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_FWD)};                 \
                sbl_launch_gemm<ConvGatherKC<BM, false>, DenseKC<BN, true>, EpiStore<0, false>, BM, BN, 1, WN>(al, bl, e, M, N, K, 1, s, sc); \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
    if (KH == 3 && stride == 1) {
        // layers 1 and 2: the input patch of a tile staged once in LDS for all nine taps (conv_patch.h)
        bool done;
        const PatchEpi pe{y, stats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (stats) done = sbl_launch_conv_patch<false, 1>(x, w, pe, NIMG, H, W, Cin, Cout, SBL_KID_CONV_FWD, s);
        else done = sbl_launch_conv_patch<false, 0>(x, w, pe, NIMG, H, W, Cin, Cout, SBL_KID_CONV_FWD, s);
        if (done) {
            SBL_LAUNCH_CHECK("sbl_conv2d_fwd(patch)");
            return 0;
        }
    }
    if (conv_pm_ok(Ho, Wo, KH, stride)) {
        // position-major rows: border pixels skip their out-of-bounds taps (mfma_gemm.h, ConvGatherPM)
#define SBL_KPM_T_(P) sbl_conv_pm_kernel<ConvGatherPM<BM_, false>, DenseKCTapList<BN_>, EpiStore<0, true>, BM_, BN_, false, P>
#define SBL_KPM_F_(P) sbl_conv_pm_kernel<ConvGatherPM<BM_, false>, DenseKCTapList<BN_>, EpiStore<0, false>, BM_, BN_, false, P>
#define SBL_CONV_FWD_PM(BM, BN)                                                                                \
    do {                                                                                                       \
        constexpr int BM_ = BM, BN_ = BN;                                                                      \
        ConvGatherPM<BM, false> al{x, g, M, 0ull};                                                             \
        DenseKCTapList<BN> bl{w, (long)K, N, Cin, 0ull};                                                       \
        SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_FWD)};                         \
        dim3 grid(sbl_cdiv(N, BN), sbl_cdiv(M, BM), 1);                                                        \
        if (stats) {                                                                                           \
            EpiStore<0, true> e{y, (long)N, nullptr, 0, stats, nullptr, 0, 2, NIMG, Ho * Wo, 0, 0, 0, 0};      \
            SBL_PREC_LAUNCH(SBL_KPM_T_, grid, s, al, bl, e, sc, M, N);                                        \
        } else {                                                                                               \
            EpiStore<0, false> e{y, (long)N, nullptr, 0, nullptr, nullptr, 0, 2, NIMG, Ho * Wo, 0, 0, 0, 0};   \
            SBL_PREC_LAUNCH(SBL_KPM_F_, grid, s, al, bl, e, sc, M, N);                                        \
        }                                                                                                      \
    } while (0)
        if (g_pm_tile == 1 || (!g_pm_tile && N >= 128 && t128 >= 512)) SBL_CONV_FWD_PM(128, 128);
        else if (g_pm_tile == 2 || (!g_pm_tile && (long)sbl_cdiv(M, 128) * sbl_cdiv(N, 64) >= 512)) SBL_CONV_FWD_PM(128, 64);
        else SBL_CONV_FWD_PM(64, 64);
#undef SBL_CONV_FWD_PM
#undef SBL_KPM_T_
#undef SBL_KPM_F_
        SBL_LAUNCH_CHECK("sbl_conv2d_fwd(pm)");
        return 0;
    }
    // all tiles are co-resident (<= 4 workgroups per CU), so the launch lasts as long as the fullest CU: pick the
    // largest tile whose count per CU (256 CUs) does not round up by more than ~20 % (522 128x128 tiles = 2.04/CU
    // would run at 3/CU speed; 1044 128x64 tiles = 4.08/CU at 5/CU)
    constexpr int q128 = 1;
    const bool waste128 = q128 && N >= 128 && t128 >= 512 && t128 < 1024 && (double)(sbl_cdiv(t128, 256) * 256) / (double)t128 > 1.25;
    // (a 256x64 tile with 4x1 wavefronts of 64x64 was measured slower than 128x64 on the 64-channel layer: 442 vs 405 us)
    if (N >= 128 && t128 >= 512 && !waste128) SBL_CONV_FWD(128, 128, 2);
    else if ((N < 128 || waste128) && (long)sbl_cdiv(M, 128) * sbl_cdiv(N, 64) >= 512) SBL_CONV_FWD(128, 64, 2);
    else SBL_CONV_FWD(64, 64, 2);
#undef SBL_CONV_FWD
    SBL_LAUNCH_CHECK("sbl_conv2d_fwd");
    return 0;
}

// What may ride on an input-gradient convolution's epilogue (all optional):
//   addend      the residual branch's gradient, added before the store: laid out like dx (identity shortcut), or - stride-2
//               convolutions, add_class00 - the compact (NIMG, ceil(H/2), ceil(W/2), Cin) gradient of the 1x1 / stride-2
//               downsample branch, which lives on the even/even pixels only (the rows of parity class (0,0));
//   y,x,mean,inv the BatchNorm whose output gradient dx is (dx = d relu(bn(.)) of the producing block): its two backward
//               sums over g = dx * (y > 0);  x2,mean2,inv2: a second BatchNorm fed by the same y (that block's
//               downsample branch): sums[2C..4C) = (sum g, sum g * xhat2).
struct DgradFuse {
    const float* addend;
    int add_class00;
    const float* y;
    const float* x;
    const float* mean;
    const float* inv;
    const float* x2;
    const float* mean2;
    const float* inv2;
    double* sums;
    int sums_zeroed;       // the caller hands over zeros (one pooled memset per step instead of one per convolution)
};
static int conv2d_dgrad_impl(const float* dy, const float* wt, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, void* ws, long ws_bytes, sbl_stream_t stream,
                             const DgradFuse& f, int compact_out) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_conv("sbl_conv2d_dgrad", NIMG, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    SBL_REQUIRE(dy && wt && dx && sbl_aligned16(dy) && sbl_aligned16(wt), "sbl_conv2d_dgrad: null/unaligned pointer");
    const int Ho = out_dim(H, KH, stride, pad), Wo = out_dim(W, KW, stride, pad);
    const bool fused = f.addend || f.sums;
    if (f.sums) {
        SBL_REQUIRE(f.y && f.x && f.mean && f.inv && (!f.x2 || (f.mean2 && f.inv2)), "sbl_conv2d_dgrad_fused: incomplete BatchNorm operands");
        if (!f.sums_zeroed) SBL_HIP(hipMemsetAsync(f.sums, 0, sizeof(double) * (f.x2 ? 4 : 2) * Cin, s));
    }
    SBL_REQUIRE(!f.addend || f.add_class00 == (stride == 2), "sbl_conv2d_dgrad_fused: the compact addend belongs to stride-2 convolutions (and only to them)");
    SBL_REQUIRE(!compact_out || (KH == 1 && stride == 2 && !fused), "sbl_conv2d_dgrad: compact output is the 1x1 / stride-2 case");
    if (stride == 2) {
        // Input pixel (ih, iw) only receives taps with kh = ih + pad (mod 2), kw likewise: 1 + 2 + 2 + 4 of the 9 taps
        // over the four parity classes (3x3), or the even/even class alone (1x1).  One dense implicit GEMM per class
        // (rows = the class's pixels, k = its taps) does 1/4 of the work of gathering zeros for the other taps.
        const int N = Cin;
        if (KH == 1 && !compact_out) SBL_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)NIMG * H * W * Cin, s));
        // the classes, heaviest (most taps) first
        struct Cls { ConvGeom g; int lin[4]; int M, K, ph, pw; };
        Cls cls[4];
        int nc = 0;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                Cls c{ConvGeom{NIMG, (H - ph + 1) / 2, (W - pw + 1) / 2, Ho, Wo, Cout, KH, KW, stride, pad, 1, ph, pw, 0, {0, 0, 0, 0}, {0, 0, 0, 0}}, {0, 0, 0, 0}, 0, 0, ph, pw};
                sbl_geom_finish(c.g);
                for (int kh = 0; kh < KH; ++kh)
                    for (int kw = 0; kw < KW; ++kw)
                        if (((ph + pad - kh) & 1) == 0 && ((pw + pad - kw) & 1) == 0) {
                            c.g.tkh[c.g.ntaps] = kh; c.g.tkw[c.g.ntaps] = kw; c.lin[c.g.ntaps] = kh * KW + kw;
                            ++c.g.ntaps;
                        }
                c.M = NIMG * c.g.OH * c.g.OW;
                c.K = c.g.ntaps * Cout;
                if (c.g.ntaps == 0 || c.M == 0) continue;
                int at = nc++;
                while (at > 0 && cls[at - 1].K < c.K) { cls[at] = cls[at - 1]; --at; }
                cls[at] = c;
            }
        if (nc == 0) return 0;
        SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_DGRAD)};
        const int cmap = compact_out ? 0 : 1;
        long t128 = 0, t128x64 = 0;
        for (int i = 0; i < nc; ++i) {
            t128 += (long)sbl_cdiv(cls[i].M, 128) * sbl_cdiv(N, 128);
            t128x64 += (long)sbl_cdiv(cls[i].M, 128) * sbl_cdiv(N, 64);
        }
#define SBL_CONV_DGC_GO(BM, BN, EPI_T, MAKE_EPI)                                                              \
    do {                                                                                                      \
        ClassSet<ConvGatherKC<BM, true>, DenseKCTaps<BN>, EPI_T> cs;                                          \
        cs.nclass = nc;                                                                                       \
        cs.tiles_n = sbl_cdiv(N, BN);                                                                         \
        long tt = 0;                                                                                          \
        for (int i = 0; i < SBL_MAX_CLASSES; ++i) {                                                           \
            const Cls& c = cls[i < nc ? i : 0];                                                               \
            const float* add = (f.addend && c.ph == 0 && c.pw == 0) ? f.addend : nullptr;                     \
            cs.al[i] = ConvGatherKC<BM, true>{dy, c.g, c.M};                                                  \
            cs.bl[i] = DenseKCTaps<BN>{wt, (long)KH * KW * Cout, N, Cout, {c.lin[0], c.lin[1], c.lin[2], c.lin[3]}}; \
            cs.epi[i] = MAKE_EPI;                                                                             \
            cs.M[i] = c.M; cs.K[i] = c.K; cs.t0[i] = (int)tt;                                                 \
            if (i < nc) tt += (long)sbl_cdiv(c.M, BM) * cs.tiles_n;                                           \
        }                                                                                                     \
        cs.t0[SBL_MAX_CLASSES] = (int)tt;                                                                     \
        SBL_REQUIRE(tt < (1L << 30), "sbl_conv2d_dgrad: too many tiles");                                     \
        SBL_PREC_LAUNCH(SBL_KCL_, dim3((unsigned)tt), s, cs, sc, N);                                          \
    } while (0)
#define SBL_CONV_DGC(BM, BN)                                                                                  \
    do {                                                                                                      \
        constexpr int BM_ = BM, BN_ = BN;                                                                     \
        if (f.sums) {                                                                                         \
            using E_ = EpiStore<0, true, true>;                                                               \
            SBL_CONV_DGC_GO(BM, BN, E_, (E_{dx, (long)N, nullptr, 0, f.sums, nullptr, 0, cmap, c.g.OH, c.g.OW, H, W, c.ph, c.pw, \
                                            f.y, f.x, f.mean, f.inv, f.x2, f.mean2, f.inv2, add, (long)N}));   \
        } else if (fused) {                                                                                   \
            using E_ = EpiStore<0, false, true>;                                                              \
            SBL_CONV_DGC_GO(BM, BN, E_, (E_{dx, (long)N, nullptr, 0, nullptr, nullptr, 0, cmap, c.g.OH, c.g.OW, H, W, c.ph, c.pw, \
                                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, add, (long)N})); \
        } else {                                                                                              \
            using E_ = EpiStore<0, false>;                                                                    \
            SBL_CONV_DGC_GO(BM, BN, E_, (E_{dx, (long)N, nullptr, 0, nullptr, nullptr, 0, cmap, c.g.OH, c.g.OW, H, W, c.ph, c.pw})); \
        }                                                                                                     \
    } while (0)
#define SBL_KCL_(P) sbl_conv_classes_kernel<ConvGatherKC<BM_, true>, DenseKCTaps<BN_>, E_, BM_, BN_, P>
        // (the 128x128 tile with the fused statistics epilogue needs > 168 registers: one wave per SIMD; 128x64 instead)
        if (N >= 128 && t128 >= 512 && !f.sums) SBL_CONV_DGC(128, 128);
        else if ((N < 128 || f.sums) && t128x64 >= 512) SBL_CONV_DGC(128, 64);
        else SBL_CONV_DGC(64, 64);
#undef SBL_KCL_
#undef SBL_CONV_DGC
#undef SBL_CONV_DGC_GO
        SBL_LAUNCH_CHECK("sbl_conv2d_dgrad(classes)");
        return 0;
    }
    const int M = NIMG * H * W, N = Cin, K = KH * KW * Cout;
    ConvGeom g{NIMG, H, W, Ho, Wo, Cout, KH, KW, stride, pad, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
    sbl_geom_finish(g);
    const long t128 = (long)sbl_cdiv(M, 128) * sbl_cdiv(N, 128);
    SBL_REQUIRE(!ws || (sbl_aligned16(ws) && ws_bytes >= (long)sizeof(int) * SBL_CONV_WS_COUNTERS), "sbl_conv2d_dgrad: workspace unaligned or < 16 KiB");
    if (KH == 3 && stride == 1) {
        // large maps: patch-resident kernel with mirrored taps (conv_patch.h); same epilogue functors
        bool done;
        const PatchEpi pe{dx, f.sums, f.addend, f.y, f.x, f.mean, f.inv, f.x2, f.mean2, f.inv2};
        if (f.sums) done = sbl_launch_conv_patch<true, 2>(dy, wt, pe, NIMG, H, W, Cout, Cin, SBL_KID_CONV_DGRAD, s);
        else if (f.addend) done = sbl_launch_conv_patch<true, 3>(dy, wt, pe, NIMG, H, W, Cout, Cin, SBL_KID_CONV_DGRAD, s);
        else done = sbl_launch_conv_patch<true, 0>(dy, wt, pe, NIMG, H, W, Cout, Cin, SBL_KID_CONV_DGRAD, s);
        if (done) {
            SBL_LAUNCH_CHECK("sbl_conv2d_dgrad(patch)");
            return 0;
        }
    }
    if (conv_pm_ok(H, W, KH, stride)) {
#define SBL_KPM_T_(P) sbl_conv_pm_kernel<ConvGatherPM<BM_, true>, DenseKCTapList<BN_>, EpiStore<0, true, true>, BM_, BN_, true, P>
#define SBL_KPM_A_(P) sbl_conv_pm_kernel<ConvGatherPM<BM_, true>, DenseKCTapList<BN_>, EpiStore<0, false, true>, BM_, BN_, true, P>
#define SBL_KPM_F_(P) sbl_conv_pm_kernel<ConvGatherPM<BM_, true>, DenseKCTapList<BN_>, EpiStore<0, false>, BM_, BN_, true, P>
#define SBL_CONV_DG_PM(BM, BN)                                                                                 \
    do {                                                                                                       \
        constexpr int BM_ = BM, BN_ = BN;                                                                      \
        ConvGatherPM<BM, true> al{dy, g, M, 0ull};                                                             \
        DenseKCTapList<BN> bl{wt, (long)K, N, Cout, 0ull};                                                     \
        SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_DGRAD)};                       \
        dim3 grid(sbl_cdiv(N, BN), sbl_cdiv(M, BM), 1);                                                        \
        if (f.sums) {                                                                                          \
            EpiStore<0, true, true> e{dx, (long)N, nullptr, 0, f.sums, nullptr, 0, 2, NIMG, H * W, 0, 0, 0, 0, \
                                      f.y, f.x, f.mean, f.inv, f.x2, f.mean2, f.inv2, f.addend, 0L};           \
            SBL_PREC_LAUNCH(SBL_KPM_T_, grid, s, al, bl, e, sc, M, N);                                        \
        } else if (f.addend) {                                                                                 \
            EpiStore<0, false, true> e{dx, (long)N, nullptr, 0, nullptr, nullptr, 0, 2, NIMG, H * W, 0, 0, 0, 0, \
                                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, f.addend, 0L}; \
            SBL_PREC_LAUNCH(SBL_KPM_A_, grid, s, al, bl, e, sc, M, N);                                        \
        } else {                                                                                               \
            EpiStore<0, false> e{dx, (long)N, nullptr, 0, nullptr, nullptr, 0, 2, NIMG, H * W, 0, 0, 0, 0};    \
            SBL_PREC_LAUNCH(SBL_KPM_F_, grid, s, al, bl, e, sc, M, N);                                        \
        }                                                                                                      \
    } while (0)
        if (g_pm_tile == 1 || (!g_pm_tile && N >= 128 && t128 >= 512)) SBL_CONV_DG_PM(128, 128);
        else if (g_pm_tile == 2 || (!g_pm_tile && (long)sbl_cdiv(M, 128) * sbl_cdiv(N, 64) >= 512)) SBL_CONV_DG_PM(128, 64);
        else SBL_CONV_DG_PM(64, 64);
#undef SBL_CONV_DG_PM
#undef SBL_KPM_T_
#undef SBL_KPM_A_
#undef SBL_KPM_F_
        SBL_LAUNCH_CHECK("sbl_conv2d_dgrad(pm)");
        return 0;
    }
#define SBL_CONV_DG(BM, BN, WN)                                                                               \
    do {                                                                                                      \
        ConvGatherKC<BM, true> al{dy, g, M};                                                                  \
        DenseKC<BN, true> bl{wt, (long)K, N};                                                                 \
        if (f.sums) {                                                                                         \
            EpiStore<0, true, true> e{dx, (long)N, nullptr, 0, f.sums, nullptr, 0, 0, 0, 0, 0, 0, 0, 0,       \
                                      f.y, f.x, f.mean, f.inv, f.x2, f.mean2, f.inv2, f.addend, 0L};          \
            if (!(g_tailsplit && sbl_launch_gemm_tailsplit<ConvGatherKC<BM, true>, DenseKC<BN, true>, EpiStore<0, true, true>, BM, BN, 1>(al, bl, e, M, N, K, s, SBL_KID_CONV_DGRAD, ws, ws_bytes, SBL_CONV_WS_COUNTERS))) { \
                SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_DGRAD)};              \
                sbl_launch_gemm<ConvGatherKC<BM, true>, DenseKC<BN, true>, EpiStore<0, true, true>, BM, BN, 1, WN>(al, bl, e, M, N, K, 1, s, sc); \
            }                                                                                                 \
        } else if (f.addend) {                                                                                \
            EpiStore<0, false, true> e{dx, (long)N, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0,     \
                                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, f.addend, 0L}; \
            if (!(g_tailsplit && sbl_launch_gemm_tailsplit<ConvGatherKC<BM, true>, DenseKC<BN, true>, EpiStore<0, false, true>, BM, BN, 1>(al, bl, e, M, N, K, s, SBL_KID_CONV_DGRAD, ws, ws_bytes, SBL_CONV_WS_COUNTERS))) { \
                SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_DGRAD)};              \
                sbl_launch_gemm<ConvGatherKC<BM, true>, DenseKC<BN, true>, EpiStore<0, false, true>, BM, BN, 1, WN>(al, bl, e, M, N, K, 1, s, sc); \
            }                                                                                                 \
        } else {                                                                                              \
            EpiStore<0, false> e{dx, (long)N, nullptr, 0, nullptr, nullptr, 0};                               \
            if (!(g_tailsplit && sbl_launch_gemm_tailsplit<ConvGatherKC<BM, true>, DenseKC<BN, true>, EpiStore<0, false>, BM, BN, 1>(al, bl, e, M, N, K, s, SBL_KID_CONV_DGRAD, ws, ws_bytes, SBL_CONV_WS_COUNTERS))) { \
                SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_DGRAD)};              \
                sbl_launch_gemm<ConvGatherKC<BM, true>, DenseKC<BN, true>, EpiStore<0, false>, BM, BN, 1, WN>(al, bl, e, M, N, K, 1, s, sc); \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
    constexpr int q128 = 1;
    const bool waste128 = q128 && N >= 128 && t128 >= 512 && t128 < 1024 && (double)(sbl_cdiv(t128, 256) * 256) / (double)t128 > 1.25;
    if (N >= 128 && t128 >= 512 && !waste128 && !f.sums) SBL_CONV_DG(128, 128, 2);
    else if ((N < 128 || waste128 || f.sums) && (long)sbl_cdiv(M, 128) * sbl_cdiv(N, 64) >= 512) SBL_CONV_DG(128, 64, 2);
    else SBL_CONV_DG(64, 64, 2);
#undef SBL_CONV_DG
    SBL_LAUNCH_CHECK("sbl_conv2d_dgrad");
    return 0;
}
static const DgradFuse kNoFuse{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
extern "C" int sbl_conv2d_dgrad(const float* dy, const float* wt, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad, void* ws, long ws_bytes, sbl_stream_t stream) {
    return conv2d_dgrad_impl(dy, wt, dx, NIMG, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes, stream, kNoFuse, 0);
}
extern "C" int sbl_conv2d_dgrad_bnstats(const float* dy, const float* wt, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                                        int KH, int KW, int stride, int pad, void* ws, long ws_bytes, const float* act,
                                        const float* pre, const float* mean, const float* invstd, double* sums,
                                        int sums_zeroed, sbl_stream_t stream) {
    SBL_REQUIRE(act && pre && mean && invstd && sums, "sbl_conv2d_dgrad_bnstats: null statistics operand");
    return conv2d_dgrad_impl(dy, wt, dx, NIMG, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes, stream,
                             DgradFuse{nullptr, 0, act, pre, mean, invstd, nullptr, nullptr, nullptr, sums, sums_zeroed}, 0);
}
extern "C" int sbl_conv2d_dgrad_fused(const float* dy, const float* wt, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                                      int KH, int KW, int stride, int pad, void* ws, long ws_bytes, const float* addend,
                                      const float* act, const float* pre, const float* mean, const float* invstd,
                                      const float* pre2, const float* mean2, const float* invstd2, double* sums,
                                      int sums_zeroed, sbl_stream_t stream) {
    return conv2d_dgrad_impl(dy, wt, dx, NIMG, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes, stream,
                             DgradFuse{addend, addend && stride == 2, act, pre, mean, invstd, pre2, mean2, invstd2, sums, sums_zeroed}, 0);
}
extern "C" int sbl_conv1x1s2_dgrad_compact(const float* dy, const float* wt, float* dx_compact, int NIMG, int H, int W, int Cin,
                                           int Cout, void* ws, long ws_bytes, sbl_stream_t stream) {
    return conv2d_dgrad_impl(dy, wt, dx_compact, NIMG, H, W, Cin, Cout, 1, 1, 2, 0, ws, ws_bytes, stream, kNoFuse, 1);
}

extern "C" int sbl_conv2d_wgrad(const float* x, const float* dy, float* dw, int NIMG, int H, int W, int Cin, int Cout,
                                int KH, int KW, int stride, int pad, int dw_zeroed, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_conv("sbl_conv2d_wgrad", NIMG, H, W, Cin, Cout, KH, KW, stride, pad)) return e;
    SBL_REQUIRE(x && dy && dw && sbl_aligned16(x) && sbl_aligned16(dy), "sbl_conv2d_wgrad: null/unaligned pointer");
    const int Ho = out_dim(H, KH, stride, pad), Wo = out_dim(W, KW, stride, pad);
    const int M = Cout, N = KH * KW * Cin, K = NIMG * Ho * Wo;   // reduce over output pixels
    ConvGeom g{NIMG, Ho, Wo, H, W, Cin, KH, KW, stride, pad, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
    sbl_geom_finish(g);
    if (!dw_zeroed) SBL_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)M * N, s));
    // split the pixel reduction: 128x128 tiles for the 128+-channel layers (twice the flops per staged byte), 64x64
    // otherwise, and 6 / 12 workgroups per CU so that the uneven last chunks and the atomic epilogues of one
    // workgroup hide behind the others (measured, tools/bench_conv.py: 465/477/511/515 us -> 411/355/437/453 us for
    // layers 1-4); chunks stay >= 256 pixels
    constexpr int wg_tile = 0;
    constexpr int wg_target_env = 0;
    const bool big = wg_tile ? wg_tile == 128 : (M >= 128 && N >= 1152 && !(stride == 2 && g_sbl_wg_s2_small));
    const int wg_target = wg_target_env ? wg_target_env : (g_sbl_wg_target > 0 && stride == 2 ? g_sbl_wg_target : (big ? 1536 : 3072));
    SplitCtl sc{nullptr, nullptr, nullptr, sbl_next_stamp_slot(SBL_KID_CONV_WGRAD)};
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && sbl_launch_conv_patch_wgrad(x, dy, dw, NIMG, H, W, Cin, Cout, sc.stamp, s)) {
        SBL_LAUNCH_CHECK("sbl_conv2d_wgrad(patch)");
        return 0;
    }
#define SBL_CONV_WG(BM, BN)                                                                                   \
    do {                                                                                                      \
        const long tiles = (long)sbl_cdiv(M, BM) * sbl_cdiv(N, BN);                                           \
        int splits = (int)((wg_target + tiles - 1) / tiles);                                                  \
        if (splits > K / 256) splits = K / 256;                                                               \
        if (splits < 1) splits = 1;                                                                           \
        DenseMC<BM, true> al{dy, (long)Cout, M};                                                              \
        ConvGatherMC<BN> bl{x, g, N};                                                                         \
        EpiStore<2, false> e{dw, (long)N, nullptr, 0, nullptr, nullptr, 0};                                   \
        sbl_launch_gemm<DenseMC<BM, true>, ConvGatherMC<BN>, EpiStore<2, false>, BM, BN>(al, bl, e, M, N, K, splits, s, sc); \
    } while (0)
    const int pm_wg_tile = (g_sbl_pm_wg64_maxm > 0 && M <= g_sbl_pm_wg64_maxm) ? 64 : 128;
    if (conv_pm_ok(Ho, Wo, KH, stride) && big && M >= 128 && Cin % 128 == 0) {
        // one tap per tile of the (tap, ci) axis: contract only over the pixels that tap can reach
#define SBL_KPMW_(P) sbl_conv_pm_wgrad_kernel<DenseMCPM<T_>, ConvGatherMCPM<T_>, EpiStore<2, false>, T_, T_, P>
#define SBL_CONV_WG_PM(T)                                                                                     \
    do {                                                                                                      \
        constexpr int T_ = T;                                                                                 \
        const long tiles = (long)sbl_cdiv(M, T) * sbl_cdiv(N, T);                                             \
        const int target = (T == 128) ? wg_target : 2 * wg_target;                                            \
        int splits = (int)((target + tiles - 1) / tiles);                                                     \
        if (splits > K / 256) splits = K / 256;                                                               \
        if (splits < 1) splits = 1;                                                                           \
        DenseMCPM<T> al{dy, (long)Cout, M, NIMG, Ho, Wo, g.fdNIMG, PmRect{0, 0, 1, 0, 0, 1.f}};                              \
        ConvGatherMCPM<T> bl{x, g, N, PmRect{0, 0, 1, 0, 0, 1.f}};                                              \
        EpiStore<2, false> e{dw, (long)N, nullptr, 0, nullptr, nullptr, 0};                                   \
        SBL_PREC_LAUNCH(SBL_KPMW_, dim3(sbl_cdiv(M, T), sbl_cdiv(N, T), splits), s, al, bl, e, sc, M, N);     \
    } while (0)
        if (pm_wg_tile == 64) SBL_CONV_WG_PM(64);
        else SBL_CONV_WG_PM(128);
#undef SBL_CONV_WG_PM
#undef SBL_KPMW_
        SBL_LAUNCH_CHECK("sbl_conv2d_wgrad(pm)");
        return 0;
    }
    if (big && M >= 128) SBL_CONV_WG(128, 128);
    else SBL_CONV_WG(64, 64);
#undef SBL_CONV_WG
    SBL_LAUNCH_CHECK("sbl_conv2d_wgrad");
    return 0;
}

// ------------------------------------------------------------------ weight layout
// OIHW (state-dict layout) -> OHWI [Cout][KH][KW][Cin]  (+ dgrad operand [Cin][KH][KW][Cout])
__global__ void weight_pack_kernel(const float* __restrict__ w, float* __restrict__ ohwi, float* __restrict__ wt,
                                   int Cout, int Cin, int KH, int KW, double* __restrict__ zero, int nzero) {
    long n = (long)Cout * Cin * KH * KW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x) zero[i] = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // i indexes OHWI
        int ci = i % Cin;
        long t = i / Cin;
        int kw = t % KW;
        t /= KW;
        int kh = t % KH;
        int co = t / KH;
        float v = w[(((long)co * Cin + ci) * KH + kh) * KW + kw];
        ohwi[i] = v;
        if (wt) wt[(((long)ci * KH + kh) * KW + kw) * Cout + co] = v;
    }
}
__global__ void wgrad_unpack_kernel(const float* __restrict__ ohwi, float* __restrict__ oihw, int Cout, int Cin, int KH,
                                    int KW, int accumulate) {
    long n = (long)Cout * Cin * KH * KW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // i indexes OIHW
        int kw = i % KW;
        long t = i / KW;
        int kh = t % KH;
        t /= KH;
        int ci = t % Cin;
        int co = t / Cin;
        const float g = ohwi[(((long)co * KH + kh) * KW + kw) * Cin + ci];
        oihw[i] = accumulate ? oihw[i] + g : g;
    }
}
extern "C" int sbl_conv_weight_pack(const float* w, float* ohwi, float* wt, int Cout, int Cin, int KH, int KW,
                                    double* zero, int nzero, sbl_stream_t stream) {
    SBL_REQUIRE(w && ohwi && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && nzero >= 0 && (zero || !nzero), "sbl_conv_weight_pack: bad args");
    long n = (long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(weight_pack_kernel, dim3(sbl_cdiv(n, 256) > 2048 ? 2048 : sbl_cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, ohwi, wt, Cout, Cin, KH, KW, zero, nzero);
    SBL_LAUNCH_CHECK("sbl_conv_weight_pack");
    return 0;
}
extern "C" int sbl_conv_wgrad_unpack(const float* ohwi, float* oihw, int Cout, int Cin, int KH, int KW, int accumulate,
                                     sbl_stream_t stream) {
    SBL_REQUIRE(ohwi && oihw && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "sbl_conv_wgrad_unpack: bad args");
    long n = (long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(sbl_cdiv(n, 256) > 2048 ? 2048 : sbl_cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, ohwi, oihw, Cout, Cin, KH, KW, accumulate);
    SBL_LAUNCH_CHECK("sbl_conv_wgrad_unpack");
    return 0;
}

// ------------------------------------------------------------------ global average pool (NHWC)
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NIMG, int HW, int C) {
    long n = (long)NIMG * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = i % C;
        long img = i / C;
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += x[(img * HW + p) * C + c];
        y[i] = s / (float)HW;
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NIMG, int HW, int C) {
    long n = (long)NIMG * HW * C;
    const float inv = 1.f / (float)HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = i % C;
        long img = i / ((long)HW * C);
        dx[i] = dy[img * C + c] * inv;
    }
}
extern "C" int sbl_avgpool_fwd(const float* x, float* y, int NIMG, int HW, int C, sbl_stream_t stream) {
    SBL_REQUIRE(x && y && NIMG > 0 && HW > 0 && C > 0, "sbl_avgpool_fwd: bad args");
    long n = (long)NIMG * C;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(sbl_cdiv(n, 256) > 4096 ? 4096 : sbl_cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, NIMG, HW, C);
    SBL_LAUNCH_CHECK("sbl_avgpool_fwd");
    return 0;
}
extern "C" int sbl_avgpool_bwd(const float* dy, float* dx, int NIMG, int HW, int C, sbl_stream_t stream) {
    SBL_REQUIRE(dy && dx && NIMG > 0 && HW > 0 && C > 0, "sbl_avgpool_bwd: bad args");
    long n = (long)NIMG * HW * C;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(sbl_cdiv(n, 256) > 4096 ? 4096 : sbl_cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, NIMG, HW, C);
    SBL_LAUNCH_CHECK("sbl_avgpool_bwd");
    return 0;
}
