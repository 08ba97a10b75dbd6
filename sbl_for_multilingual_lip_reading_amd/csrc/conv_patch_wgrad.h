// Patch-resident weight gradient of the 3x3 / stride-1 / pad-1 trunk convolutions (22x22 and 11x11 maps), split-bf16 modes.
//
//   dw[co][kh][kw][ci] = sum over pixels p of  dy[p][co] * x[p + (kh - 1, kw - 1)][ci]          (video_frontend.py:10-12, backward)
//
// The implicit-GEMM weight gradients of mfma_gemm.h give every tap its own tiles: each of the nine re-fetches dy and the
// shifted x and re-splits both into bf16 planes, 16 pixels per barrier interval.  Here a persistent workgroup (768 threads =
// 12 wavefronts) owns a (64 co x 64 ci) block of all nine taps and walks pixel tiles (TR rows of an image, or G whole
// images): per tile the (TR + 2) x (W + 4) x 64-channel input patch and the (pixels x 64) dy tile are split ONCE into
// pixel-major bf16 planes in LDS - the layout the data has in memory, 8-byte stores - and every MFMA operand (8 consecutive
// PIXELS of one channel: the contraction runs over pixels) comes out of two transposed reads (ds_read_b64_tr_b16), a tap
// being nothing but a row offset into the patch image.  Wavefront (cb, jb, kh) keeps the three 32x32 accumulators
// (kw = 0, 1, 2) of its (32 co, 32 ci, kh) share for the whole launch: 18 MFMAs per 8 + 24 transposed reads at bf16x6, no
// barrier inside a tile.  The next tile's operands are fetched into registers while the current one multiplies.  One pass
// of float atomics per workgroup at the end (lanes = consecutive ci: 128-byte runs).
//
// LDS image (both operands): 128-byte rows (64 channels), the two 64-byte halves of a row swapped when bit 1 of the row
// index is set: a transposed read takes 4 consecutive rows x 64 bytes per 32-lane half, which then covers all 64 banks
// once.  The patch is W + 4 wide (not W + 2): a run of pixels that wraps to the next image row then jumps by 4 patch rows
// more, which keeps "4 consecutive pixels -> 4 row indices that differ mod 4".
#pragma once
#include "mfma_gemm.h"

#define SBL_PWG_THREADS 768
#define SBL_PWG_XQ 5      // float4 per thread of the patch prefetch  (patch rows <= 5 * 768 / 16 = 240)
#define SBL_PWG_DQ 3      // float4 per thread of the dy prefetch     (tile pixels, rounded up to 16, <= 144)

__device__ __forceinline__ int pwg_off(int row, int f) {      // byte offset of the 8-byte slot of channels 4f .. 4f + 3 of image row `row`
    return row * 128 + ((((f >> 1) ^ (((row >> 1) & 1) << 2))) << 4) + ((f & 1) << 3);
}
__device__ __forceinline__ int pwg_offc(int row, int fc) {      // the same with fc = 16 (f >> 1) + 8 (f & 1) precomputed
    return ((row << 7) + fc) ^ ((row & 2) << 5);
}
__device__ __forceinline__ int pwg_div(int v, float r) { return (int)(((float)v + 0.5f) * r); }      // v / d for small v, r = 1 / d

struct PwgGeom {
    int NIMG, H, W, Cin, Cout;
    int TR, tpi, ntiles, G;      // tile = TR rows of one image (G == 1) or G whole images (TR == H)
    int PW, PH, xrows, dyrows;   // patch width W + 4, height TR + 2; rows of the two LDS images
    float rPW, rPHPW, rW, rTRW;  // reciprocals for pwg_div
};

template <int NT>
__global__ __launch_bounds__(SBL_PWG_THREADS) void sbl_conv_patch_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                             float* __restrict__ dw, PwgGeom gm, unsigned long long* stamp) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char pwg_smem[];
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    sbl_stamp_begin(stamp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 1, jb = (wave >> 1) & 1, kh = wave >> 2;
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, half = lane >> 5, l31 = lane & 31;
    const int nj = gm.Cin >> 6;
    const int cq = blockIdx.y / nj, jq = blockIdx.y - cq * nj;
    const int xplane = gm.xrows * 128, dplane = gm.dyrows * 128;
    unsigned char* xs = pwg_smem;
    unsigned char* ds = pwg_smem + NPL * xplane;
    const float* xsrc = x + jq * 64;
    const float* dsrc = dy + cq * 64;
    const int fA = 8 * cb + 4 * (g4 & 1) + p4, fB = 8 * jb + 4 * (g4 & 1) + p4;      // this lane's 8-byte slot of a row, per operand
    const int fcB = ((fB >> 1) << 4) + ((fB & 1) << 3);
    const int lrow = 8 * (g4 >> 1) + q4;                                              // its first pixel inside a 16-pixel step
    const int doff0 = pwg_off(lrow, fA), doff1 = pwg_off(lrow + 4, fA);               // dy rows: step ks adds 16 rows = 2048 bytes
    // pixel of a (full) tile -> patch row of its tap (0, 0): the same for every tile; rows past a short tile's pixels hold zeros
    // in the dy image, so whatever patch row they name only has to exist
    int* xtab = reinterpret_cast<int*>(pwg_smem + NPL * (xplane + dplane));
    {
        const int trw = gm.TR * gm.W;
        for (int pix = tid; pix < gm.dyrows; pix += SBL_PWG_THREADS) {
            const int pc = min(pix, gm.G * trw - 1);
            const int gi = pwg_div(pc, gm.rTRW), rem = pc - gi * trw;
            const int pr = pwg_div(rem, gm.rW), pcol = rem - pr * gm.W;
            xtab[pix] = (gi * gm.PH + pr) * gm.PW + pcol;
        }
    }

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 xv[SBL_PWG_XQ], dv[SBL_PWG_DQ];
    // tile -> (first image, first row, pixel count, first pixel of dy)
    auto tile_geom = [&](int tile, int& img, int& r0, int& npix, int& gcount, long& m0) {
        const int ti = tile / gm.tpi;
        img = ti * gm.G;
        r0 = (tile - ti * gm.tpi) * gm.TR;
        gcount = min(gm.G, gm.NIMG - img);
        npix = gcount * min(gm.TR, gm.H - r0) * gm.W;
        m0 = ((long)img * gm.H + r0) * gm.W;
    };
    auto fetch = [&](int tile) {
        int img, r0, npix, gcount;
        long m0;
        tile_geom(tile, img, r0, npix, gcount, m0);
#pragma unroll
        for (int u = 0; u < SBL_PWG_XQ; ++u) {
            const int q = tid + u * SBL_PWG_THREADS;
            const int row = q >> 4, f = q & 15;
            const int gi = pwg_div(row, gm.rPHPW), rp = row - gi * (gm.PH * gm.PW);
            const int prow = pwg_div(rp, gm.rPW), pcol = rp - prow * gm.PW;
            const int ih = r0 - 1 + prow, iw = pcol - 1;
            const bool ok = row < gm.xrows && gi < gcount && (unsigned)ih < (unsigned)gm.H && (unsigned)iw < (unsigned)gm.W;
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) xv[u] = *reinterpret_cast<const float4*>(xsrc + (((long)(img + gi) * gm.H + ih) * gm.W + iw) * gm.Cin + 4 * f);
        }
#pragma unroll
        for (int u = 0; u < SBL_PWG_DQ; ++u) {
            const int q = tid + u * SBL_PWG_THREADS;
            const int row = q >> 4, f = q & 15;
            dv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < npix) dv[u] = *reinterpret_cast<const float4*>(dsrc + (m0 + row) * gm.Cout + 4 * f);
        }
    };
    auto stage = [&]() {      // registers -> bf16 planes (rows past the tile's pixels: zeros, they multiply clamped x rows)
#pragma unroll
        for (int u = 0; u < SBL_PWG_XQ; ++u) {
            const int q = tid + u * SBL_PWG_THREADS;
            const int row = q >> 4, f = q & 15;
            if (row < gm.xrows) {
                uint2 pl[NPL];
                bf_split4<NPL>(xv[u], pl);
                const int off = pwg_off(row, f);
#pragma unroll
                for (int t = 0; t < NPL; ++t) *reinterpret_cast<uint2*>(xs + t * xplane + off) = pl[t];
            }
        }
#pragma unroll
        for (int u = 0; u < SBL_PWG_DQ; ++u) {
            const int q = tid + u * SBL_PWG_THREADS;
            const int row = q >> 4, f = q & 15;
            if (row < gm.dyrows) {
                uint2 pl[NPL];
                bf_split4<NPL>(dv[u], pl);
                const int off = pwg_off(row, f);
#pragma unroll
                for (int t = 0; t < NPL; ++t) *reinterpret_cast<uint2*>(ds + t * dplane + off) = pl[t];
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < gm.ntiles) fetch(tile);
    for (; tile < gm.ntiles; tile += gridDim.x) {
        int img, r0, npix, gcount;
        long m0;
        tile_geom(tile, img, r0, npix, gcount, m0);
        __syncthreads();      // every wavefront is done with the previous tile's images
        stage();
        __syncthreads();
        if (tile + (int)gridDim.x < gm.ntiles) fetch(tile + gridDim.x);      // in flight under this tile's MFMAs
        const int nks = (npix + 15) >> 4;
        for (int ks = 0; ks < nks; ++ks) {
            // this lane's two pixels of the step (rows 16 ks + 8 (g >> 1) + q, + 4): dy rows directly, patch rows through the table
            const int xr0 = xtab[ks * 16 + lrow] + kh * gm.PW, xr1 = xtab[ks * 16 + lrow + 4] + kh * gm.PW;
            bf16x8 a[NPL];
#pragma unroll
            for (int t = 0; t < NPL; ++t) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(ds + t * dplane + ks * 2048 + doff0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(ds + t * dplane + ks * 2048 + doff1));
                a[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                bf16x8 b[NPL];
                const int o0 = pwg_offc(xr0 + kw, fcB), o1 = pwg_offc(xr1 + kw, fcB);
#pragma unroll
                for (int t = 0; t < NPL; ++t) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(xs + t * xplane + o0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(xs + t * xplane + o1));
                    b[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int t = 0; t < Tm::N; ++t) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[Tm::pa(t)], b[Tm::pb(t)], acc[kw], 0, 0, 0);
            }
        }
    }
    // D layout: column l & 31 (ci), rows (r & 3) + 8 (r >> 2) + 4 half (co)
    const int ci = jq * 64 + jb * 32 + l31;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cq * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            atomicAdd(dw + ((long)co * 9 + kh * 3 + kw) * gm.Cin + ci, acc[kw][r]);
        }
    sbl_stamp_end(stamp);
}

// Tile choice.  Returns false when the map does not take this path.
static inline bool sbl_conv_patch_wgrad_geom(int NIMG, int H, int W, int Cin, int Cout, int npl, PwgGeom& gm) {
    const int max_xrows = SBL_PWG_XQ * SBL_PWG_THREADS / 16, max_pix = SBL_PWG_DQ * SBL_PWG_THREADS / 16 / 16 * 16;
    const int PW = W + 4;
    auto fits = [&](int tr, int g) {
        const int xr = g * (tr + 2) * PW, px = g * tr * W, dr = (px + 15) / 16 * 16;
        return xr <= max_xrows && px <= max_pix && (long)(xr + dr) * 128 * npl + dr * 4 <= 160 * 1024;
    };
    int best = 0;
    for (int tr = 1; tr <= H; ++tr)
        if (fits(tr, 1)) best = tr;
    if (best <= 0) return false;
    int TR, G = 1;
    if (best >= H) {
        TR = H;
        while (fits(H, G + 1)) ++G;
    } else {
        const int parts = sbl_cdiv(H, best);
        TR = sbl_cdiv(H, parts);
    }
    if (G * TR * W < 80) return false;      // too few pixels per staged patch
    gm.NIMG = NIMG; gm.H = H; gm.W = W; gm.Cin = Cin; gm.Cout = Cout;
    gm.TR = TR; gm.G = G;
    gm.tpi = sbl_cdiv(H, TR);
    gm.ntiles = G > 1 ? sbl_cdiv(NIMG, G) : NIMG * gm.tpi;
    gm.PW = PW; gm.PH = TR + 2;
    gm.xrows = G * gm.PH * PW;
    gm.dyrows = (G * TR * W + 15) / 16 * 16;
    gm.rPW = 1.0f / (float)PW;
    gm.rPHPW = 1.0f / (float)(gm.PH * PW);
    gm.rW = 1.0f / (float)W;
    gm.rTRW = 1.0f / (float)(TR * W);
    return true;
}

extern int g_sbl_conv_patch_wgrad;      // sbl_set_tuning knob 9
// dw must be zeroed (or hold the sum to add to).  Returns false when nothing was launched.
static inline bool sbl_launch_conv_patch_wgrad(const float* x, const float* dy, float* dw, int NIMG, int H, int W, int Cin, int Cout,
                                               unsigned long long* stamp, hipStream_t s) {
    if (!g_sbl_conv_patch_wgrad || g_sbl_prec == 0 || Cin % 64 != 0 || Cout % 64 != 0 || H * W < g_sbl_conv_patch_wgrad) return false;
    const int npl = g_sbl_prec == 6 ? 3 : g_sbl_prec == 3 ? 2 : 1;
    PwgGeom gm;
    if (!sbl_conv_patch_wgrad_geom(NIMG, H, W, Cin, Cout, npl, gm)) return false;
    const int combos = (Cin / 64) * (Cout / 64);
    extern int g_sbl_exp[8];
    const int cus = g_sbl_exp[0] > 0 ? g_sbl_exp[0] : 256;      // (experiment knob 100: leave CUs to the other stream)
    int gx = cus / combos;      // one workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > gm.ntiles) gx = gm.ntiles;
    const size_t lds = (size_t)npl * (gm.xrows + gm.dyrows) * 128 + (size_t)gm.dyrows * 4;      // images + pixel -> patch row table
#define SBL_PWG_GO(P)                                                                                                          \
    do {                                                                                                                       \
        static bool set_##P[64] = {false};                                                                                     \
        int dev = 0;                                                                                                           \
        if (hipGetDevice(&dev) != hipSuccess) return false;                                                                    \
        if (!set_##P[dev & 63]) {                                                                                              \
            if (hipFuncSetAttribute((const void*)sbl_conv_patch_wgrad_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false; \
            set_##P[dev & 63] = true;                                                                                          \
        }                                                                                                                      \
        hipLaunchKernelGGL((sbl_conv_patch_wgrad_kernel<P>), dim3(gx, combos), dim3(SBL_PWG_THREADS), lds, s, x, dy, dw, gm, stamp); \
    } while (0)
    if (g_sbl_prec == 6) SBL_PWG_GO(6);
    else if (g_sbl_prec == 3) SBL_PWG_GO(3);
    else SBL_PWG_GO(1);
#undef SBL_PWG_GO
    return true;
}
