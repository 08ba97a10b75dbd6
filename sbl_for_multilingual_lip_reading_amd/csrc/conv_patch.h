// Patch-resident 3x3 / stride-1 convolution for the large trunk maps (ResNet layer 1: 22x22 x 64 channels, tiles of 11 rows;
// layer 2: 11x11 x 128 channels, tiles of two whole images), split-bf16 modes.
//
// The implicit-GEMM kernels of mfma_gemm.h treat every tap as its own K range: each of the nine taps re-gathers its operand
// from global memory and re-splits it into bf16 planes (9x the loads and conversions; 0.6-0.9 GB fetched per launch for
// < 60 MB of operands, profiles/r02_pmc_fetch_write_per_kernel.csv).  Here a workgroup owns TR output rows of one image
// (TR * W <= 256 pixels = 8 MFMA row blocks) and stages the (TR + 2) x (W + 2) x 64-channel input patch ONCE per 64-channel
// chunk: fp32 -> bf16 planes on the way into LDS (one split per element), zero halo included.  The K loop then walks
// (tap, 16-channel step): a lane's A operand (pixel l & 31 of a row block, 8 channels) is one 16-byte LDS read per plane at
// the tap's offset - no global gather, no conversion, no barrier per slab.  The B operand is the tap's 64 x 64 weight block,
// split once per workgroup into LDS planes with the same padded row layout (fetched into registers during the previous
// tap, two barriers per tap).  (A first version read the B operand straight from the OHWI weights, 32 bytes per lane at a
// 2304-byte stride: 64 cache lines per wave-load kept the address unit busy longer than the MFMAs - forward 366 us
// against 242 us for the per-tap gather kernel.)  4 wavefronts, each 2 pixel blocks x 2 channel blocks of 32x32
// accumulators (64 output channels per workgroup), 24 MFMAs per 12 16-byte LDS reads at bf16x6.
//
// Forward and input gradient are the same kernel: the gradient of a 3x3 / pad-1 / stride-1 convolution is a 3x3 / pad-1
// convolution of dy with the taps mirrored (tap (kh, kw) reads pixel (oh + 1 - kh, ow + 1 - kw)) and the [Cin][kh][kw][Cout]
// weight image, so DGRAD only flips the tap offsets.  Epilogues are the engine's EpiStore functors (BN statistics, fused
// residual / BatchNorm-backward sums), fed with the same (row block, column) accumulator tiles.
#pragma once
#include "mfma_gemm.h"

// Two LDS layouts of a patch pixel / weight row (per plane):
//   SW = false: 64 channels + 16 bytes of padding (144 B stride: conflict-free 16-byte reads); 162 KB at 11 x 22 pixels, ONE
//               workgroup per CU.
//   SW = true : 32 channels, no padding (64 B), the four 16-byte chunks of a row XOR-swizzled with bits 2-3 of the row index
//               (consecutive rows then spread over all 64 banks): 72 KB, TWO workgroups per CU, so that one workgroup's patch
//               staging and epilogue run under the other's MFMAs; twice the staging passes and barriers per 64 channels.
#define SBL_CP_CKV(SW) ((SW) ? 32 : 64)
#define SBL_CP_PIXBV(SW) ((SW) ? 64 : 144)
template <bool SW>
__device__ __forceinline__ int cp_addr(int row, int chunk) {      // byte offset of 16-byte chunk `chunk` of row `row` inside a plane
    return SW ? row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4) : row * 144 + (chunk << 4);
}
#define SBL_CP_EROW 68       // floats per pixel row of the epilogue's LDS image (64 channels + 4: conflict-free column writes)

// What the epilogue does with a finished (pixels x 64 channels) tile.  The accumulators are transposed through LDS so that
// every global access of the epilogue is a 16-byte access of 16 lanes per pixel (the engine's EpiStore works on the MFMA
// layout: 4-byte accesses, 32 lanes per row - fine beside other resident workgroups, but this kernel runs one workgroup per CU
// and nothing hides it: 12.5 of 41 us per tile in the first version).
//   STATS 0: store.   STATS 1: store + per-channel sum / sum of squares (training BatchNorm statistics, video_frontend.py:31-37).
//   STATS 2: v += addend (optional), store, and the two backward sums of the BatchNorm whose output gradient v is:
//            g = v * (bs_y > 0), sums (g, g * xhat) - plus (g, g * xhat2) of a second BatchNorm on the same activation (bs_x2).
//   STATS 3: v += addend, store (a residual gradient without statistics: the first block of layer 1).
struct PatchEpi {
    float* out;            // (NIMG*H*W, Nout) row-major
    double* stats;         // [2 * Nout] (+ [2 * Nout] for bs_x2)
    const float* add_src;  // laid out like out, or nullptr
    const float* bs_y;
    const float* bs_x;
    const float* bs_mean;
    const float* bs_inv;
    const float* bs_x2;
    const float* bs_mean2;
    const float* bs_inv2;
};

template <int NT, bool DGRAD, int STATS, bool SW>
__global__ __launch_bounds__(256) void sbl_conv_patch_kernel(const float* __restrict__ src, const float* __restrict__ wk, PatchEpi epi,
                                                             int NIMG, int H, int W, int C, int Nout, int TR, int tpi, int ntiles, int G,
                                                             unsigned long long* stamp) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    constexpr int CK = SBL_CP_CKV(SW), PIXB = SBL_CP_PIXBV(SW), Q4 = CK / 4, WQ = CK / 16;      // chunk channels, row bytes, float4s per pixel, float4s of a weight row per thread
    constexpr int WPLANE = 64 * PIXB;      // one plane of a tap's 64 x CK weight block, rows padded like patch pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
    sbl_stamp_begin(stamp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int PW = W + 2, PH = TR + 2;
    const int plane = G * PH * PW * PIXB;      // G > 1: the tile is G whole images (TR == H), each with its own halo
    unsigned char* wsm = cp_smem + NPL * plane;      // [plane][co 0..63][64 ci] of the current tap
    const int n0 = blockIdx.y * 64;
    const long M = (long)NIMG * H * W;
    // epilogue role: pixel (tid >> 4) + 16 k of the tile, channels ec4 .. ec4 + 3; per-channel sums kept over the workgroup's tiles
    const int ec4 = (tid & 15) * 4;
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f}, st3[4] = {0.f, 0.f, 0.f, 0.f};
    float4 emu = make_float4(0.f, 0.f, 0.f, 0.f), eis = emu, emu2 = emu, eis2 = emu;
    if (STATS == 2) {
        emu = *reinterpret_cast<const float4*>(epi.bs_mean + n0 + ec4);
        eis = *reinterpret_cast<const float4*>(epi.bs_inv + n0 + ec4);
        if (epi.bs_x2) {
            emu2 = *reinterpret_cast<const float4*>(epi.bs_mean2 + n0 + ec4);
            eis2 = *reinterpret_cast<const float4*>(epi.bs_inv2 + n0 + ec4);
        }
    }
    // this thread's share of a tap's weight block: output channel tid >> 2, CK / 4 input channels starting at (tid & 3) * CK / 4
    const float* wsrc = wk + ((long)(n0 + (tid >> 2)) * 9) * C + (tid & 3) * (CK / 4);
    // (a thread's WQ float4s of a weight row = chunks (tid & 3) * WQ / 2 ... : WQ = 4 -> two chunks, WQ = 2 -> one chunk)
    const int wrow = tid >> 2;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = (tile / tpi) * G, r0 = (tile - (tile / tpi) * tpi) * TR;
        const int rows = min(TR, H - r0), gcount = min(G, NIMG - img), npix = gcount * rows * W;
        const long m0 = ((long)img * H + r0) * W;
        // this lane's two A-operand pixels (clamped: rows past the tile are computed and never stored)
        int arow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = min((wave * 2 + i) * 32 + l31, npix - 1);
            const int gi = p / (TR * W), rem = p - gi * (TR * W);      // (G == 1: gi = 0)
            const int pr = rem / W, pc = rem - pr * W;
            arow[i] = (gi * PH + pr) * PW + pc;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int c0 = 0; c0 < C; c0 += CK) {
            float4 wreg[WQ];
#pragma unroll
            for (int u = 0; u < WQ; ++u) wreg[u] = *reinterpret_cast<const float4*>(wsrc + c0 + u * 4);      // tap 0
            __syncthreads();      // every wave is done reading the previous patch and weight block
            // ---- stage the patch chunk: (PH x PW) pixels x 64 channels, 16 float4 per pixel, zero halo; all of a thread's loads
            // are in flight before the first split
            const int nq = G * PH * PW * Q4;
            constexpr int QB = 10;      // float4s per thread and batch (13 x 24 pixels x 8 float4 = 2496: one batch)
            for (int q0 = tid; q0 < nq; q0 += 256 * QB) {
                float4 v[QB];
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int q = q0 + u * 256;
                    const int pix = q / Q4, c4 = (q % Q4) * 4;
                    const int gi = pix / (PH * PW), rp = pix - gi * (PH * PW);
                    const int prow = rp / PW, pcol = rp - prow * PW;
                    const int ih = r0 - 1 + prow, iw = pcol - 1;
                    const bool ok = q < nq && gi < gcount && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) v[u] = *reinterpret_cast<const float4*>(src + (((long)(img + gi) * H + ih) * W + iw) * C + c0 + c4);
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int q = q0 + u * 256;
                    if (q < nq) {
                        uint2 pl[NPL];
                        bf_split4<NPL>(v[u], pl);
                        const int off = cp_addr<SW>(q / Q4, (q % Q4) >> 1) + ((q % Q4) & 1) * 8;
#pragma unroll
                        for (int t = 0; t < NPL; ++t) *reinterpret_cast<uint2*>(cp_smem + t * plane + off) = pl[t];
                    }
                }
            }
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                // the tap's 64 x 64 weight block -> bf16 planes in LDS (loaded into registers during the previous tap)
#pragma unroll
                for (int u = 0; u < WQ; ++u) {
                    uint2 pl[NPL];
                    bf_split4<NPL>(wreg[u], pl);
#pragma unroll
                    for (int t = 0; t < NPL; ++t)
                        *reinterpret_cast<uint2*>(wsm + t * WPLANE + cp_addr<SW>(wrow, ((tid & 3) * WQ + u) >> 1) + (u & 1) * 8) = pl[t];
                }
                __syncthreads();      // weight block (and, for tap 0, the patch) complete
                if (tap < 8) {
#pragma unroll
                    for (int u = 0; u < WQ; ++u) wreg[u] = *reinterpret_cast<const float4*>(wsrc + (long)(tap + 1) * C + c0 + u * 4);
                }
                const int kh = tap / 3, kw = tap - kh * 3;
                const int trow = (DGRAD ? 2 - kh : kh) * PW + (DGRAD ? 2 - kw : kw);
                bf16x8 a[2][2][NPL], b[2][2][NPL];      // [register set][block][plane]
                auto frags = [&](int cs, int set) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int t = 0; t < NPL; ++t)
                            a[set][i][t] = *reinterpret_cast<const bf16x8*>(cp_smem + t * plane + cp_addr<SW>(arow[i] + trow, cs * 2 + half));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t = 0; t < NPL; ++t)
                            b[set][j][t] = *reinterpret_cast<const bf16x8*>(wsm + t * WPLANE + cp_addr<SW>(j * 32 + l31, cs * 2 + half));
                };
                frags(0, 0);
#pragma unroll
                for (int cs = 0; cs < CK / 16; ++cs) {
                    if (cs + 1 < CK / 16) frags(cs + 1, (cs + 1) & 1);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int t = 0; t < Tm::N; ++t)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs & 1][i][Tm::pa(t)], b[cs & 1][j][Tm::pb(t)], acc[i][j], 0, 0, 0);
                }
                __syncthreads();      // every wave has read this tap's weight block
            }
        }
        // ---- epilogue.  The last tap ended on a barrier: the patch area is free; accumulators -> LDS [pixel][68 floats]
        // (D layout: column l & 31 of block j, rows (r & 3) + 8 (r >> 2) + 4 half of block i), then one 16-byte access per
        // thread and pixel.  Rows past the tile (the padded rows of its last block) are dropped here.
        float* ep = reinterpret_cast<float*>(cp_smem);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pix = (wave * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    ep[pix * SBL_CP_EROW + j * 32 + l31] = acc[i][j][r];
                }
        __syncthreads();
        for (int pix = tid >> 4; pix < npix; pix += 16) {
            float4 v = *reinterpret_cast<const float4*>(ep + pix * SBL_CP_EROW + ec4);
            const long o = (m0 + pix) * Nout + n0 + ec4;
            if ((STATS == 2 || STATS == 3) && epi.add_src) {
                const float4 a4 = *reinterpret_cast<const float4*>(epi.add_src + o);
                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
            }
            *reinterpret_cast<float4*>(epi.out + o) = v;
            if (STATS == 1) {
                st1[0] += v.x; st1[1] += v.y; st1[2] += v.z; st1[3] += v.w;
                st2[0] += v.x * v.x; st2[1] += v.y * v.y; st2[2] += v.z * v.z; st2[3] += v.w * v.w;
            } else if (STATS == 2) {
                const float4 yv = *reinterpret_cast<const float4*>(epi.bs_y + o);
                const float4 xv = *reinterpret_cast<const float4*>(epi.bs_x + o);
                const float g0 = yv.x > 0.f ? v.x : 0.f, g1 = yv.y > 0.f ? v.y : 0.f, g2 = yv.z > 0.f ? v.z : 0.f, g3 = yv.w > 0.f ? v.w : 0.f;
                st1[0] += g0; st1[1] += g1; st1[2] += g2; st1[3] += g3;
                st2[0] += g0 * ((xv.x - emu.x) * eis.x); st2[1] += g1 * ((xv.y - emu.y) * eis.y);
                st2[2] += g2 * ((xv.z - emu.z) * eis.z); st2[3] += g3 * ((xv.w - emu.w) * eis.w);
                if (epi.bs_x2) {
                    const float4 x2 = *reinterpret_cast<const float4*>(epi.bs_x2 + o);
                    st3[0] += g0 * ((x2.x - emu2.x) * eis2.x); st3[1] += g1 * ((x2.y - emu2.y) * eis2.y);
                    st3[2] += g2 * ((x2.z - emu2.z) * eis2.z); st3[3] += g3 * ((x2.w - emu2.w) * eis2.w);
                }
            }
        }
    }
    if (STATS == 1 || STATS == 2) {
        // per-channel sums: 16 pixel groups x 16 channel quads -> LDS -> one double atomic per channel and sum
        __syncthreads();
        float* red = reinterpret_cast<float*>(cp_smem);      // [3][16 groups][64 channels]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[(0 * 16 + (tid >> 4)) * 64 + ec4 + q] = st1[q];
            red[(1 * 16 + (tid >> 4)) * 64 + ec4 + q] = st2[q];
            red[(2 * 16 + (tid >> 4)) * 64 + ec4 + q] = st3[q];
        }
        __syncthreads();
        if (tid < 192) {
            const int which = tid >> 6, c = tid & 63;
            float t = 0.f;
#pragma unroll
            for (int gq = 0; gq < 16; ++gq) t += red[(which * 16 + gq) * 64 + c];
            if (which == 0) {
                atomicAdd(epi.stats + n0 + c, (double)t);
                if (STATS == 2 && epi.bs_x2) atomicAdd(epi.stats + 2 * Nout + n0 + c, (double)t);
            } else if (which == 1) {
                atomicAdd(epi.stats + Nout + n0 + c, (double)t);
            } else if (STATS == 2 && epi.bs_x2) {
                atomicAdd(epi.stats + 3 * Nout + n0 + c, (double)t);
            }
        }
    }
    sbl_stamp_end(stamp);
}

// Tile of an H x W map: TR rows of one image (G = 1; the largest TR with TR * W <= 256 whose patch fits the LDS budget, split
// evenly over the image), or - small maps, whole images only - G images of H rows (G * H * W <= 256).  Returns false when
// the map does not take this path (fewer than 7 of the tile's 8 row blocks would be used, or nothing fits).
static inline bool sbl_conv_patch_tile(int H, int W, int nplanes, bool sw, int max_imgs, int& TR, int& G) {
    const long budget = (sw ? 80 : 160) * 1024;
    const int pixb = SBL_CP_PIXBV(sw);
    int best = 0;
    for (int tr = 1; tr <= H; ++tr)
        if (tr * W <= 256 && ((long)(tr + 2) * (W + 2) + 64) * pixb * nplanes <= budget) best = tr;
    if (best >= H) {      // a whole image fits: several per tile?
        int g = 256 / (H * W);
        if (max_imgs > 0 && g > max_imgs) g = max_imgs;
        while (g > 1 && ((long)g * (H + 2) * (W + 2) + 64) * pixb * nplanes > budget) --g;
        TR = H;
        G = g < 1 ? 1 : g;
        return G * H * W >= 224 || (G == 1 && H * W >= 160);
    }
    if (best <= 0) return false;
    const int parts = sbl_cdiv(H, best);      // an even split of the image (22 rows: 11 + 11)
    TR = sbl_cdiv(H, parts);
    G = 1;
    return TR >= 4 && TR * W >= 160;
}

extern int g_sbl_conv_patch, g_sbl_conv_patch_imgs;      // sbl_set_tuning knobs 5 and 8
template <bool DGRAD, int STATS>
static inline bool sbl_launch_conv_patch(const float* src, const float* wk, const PatchEpi& epi, int NIMG, int H, int W, int C, int Nout,
                                         int kid, hipStream_t s) {
    const bool sw = g_sbl_conv_patch == 2;      // knob 5: 1 = padded 64-channel rows (one workgroup per CU), 2 = swizzled 32-channel rows (two)
    if (!g_sbl_conv_patch || g_sbl_prec == 0 || C % 64 != 0 || Nout % 64 != 0) return false;
    const int npl = g_sbl_prec == 6 ? 3 : g_sbl_prec == 3 ? 2 : 1;
    int TR = 0, G = 1;
    if (!sbl_conv_patch_tile(H, W, npl, sw, g_sbl_conv_patch_imgs, TR, G)) return false;      // small maps keep the position-major kernels
    const int tpi = sbl_cdiv(H, TR), ntiles = G > 1 ? sbl_cdiv(NIMG, G) : NIMG * tpi;
    size_t lds = (size_t)npl * ((size_t)G * (TR + 2) * (W + 2) + 64) * SBL_CP_PIXBV(sw);
    if (lds < (size_t)256 * SBL_CP_EROW * 4) lds = (size_t)256 * SBL_CP_EROW * 4;      // the epilogue's image of the tile
    const int cap = sw ? 512 : 256;      // persistent: one (two) workgroup(s) per CU
    const int gy = Nout / 64, capx = cap / gy > 0 ? cap / gy : 1;
    const int gx = ntiles < capx ? ntiles : capx;
    const dim3 grid(gx, gy);
    unsigned long long* stamp = sbl_next_stamp_slot(kid);
#define SBL_CP_GO(P, S)                                                                                                        \
    do {                                                                                                                       \
        static bool set_##P##S[64] = {false};                                                                                  \
        int dev = 0;                                                                                                           \
        if (hipGetDevice(&dev) != hipSuccess) return false;                                                                    \
        if (!set_##P##S[dev & 63]) {                                                                                           \
            if (hipFuncSetAttribute((const void*)sbl_conv_patch_kernel<P, DGRAD, STATS, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false; \
            set_##P##S[dev & 63] = true;                                                                                       \
        }                                                                                                                      \
        hipLaunchKernelGGL((sbl_conv_patch_kernel<P, DGRAD, STATS, S>), grid, dim3(256), lds, s, src, wk, epi, NIMG, H, W, C, Nout, TR, tpi, ntiles, G, stamp); \
    } while (0)
    if (sw) {
        if (g_sbl_prec == 6) SBL_CP_GO(6, true);
        else if (g_sbl_prec == 3) SBL_CP_GO(3, true);
        else SBL_CP_GO(1, true);
    } else {
        if (g_sbl_prec == 6) SBL_CP_GO(6, false);
        else if (g_sbl_prec == 3) SBL_CP_GO(3, false);
        else SBL_CP_GO(1, false);
    }
#undef SBL_CP_GO
    return true;
}
