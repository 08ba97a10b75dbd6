// frontend3D stem: Conv3d(1,64,(5,7,7),s(1,2,2),p(2,3,3)) -> BatchNorm3d -> ReLU -> MaxPool3d((1,3,3),s(1,2,2),p(0,1,1))
// (SBL/transformer/video_frontend.py:99-104), forward and backward, channels-last output.
//
// conv fwd / wgrad are implicit GEMMs on v_mfma_f32_32x32x2_f32 with the (5 x 21 x 37) input patch of an
// 8x16 output-pixel tile staged once in LDS (coalesced row reads of the (T,H,W) volume) and, for the forward,
// all 64x245 weights resident in LDS for the lifetime of a persistent workgroup.  C_in = 1, so K = 245 taps.
// Training BatchNorm needs the batch statistics before it can normalise: pass 1 writes conv_out and reduces
// per-channel sum / sum-of-squares; pass 2 normalises + ReLU + pools (and records the pool argmax).  Backward is
// two passes too: a reduction for the BN adjoint, then a weight-gradient GEMM whose A operand (dconv) is
// recomputed on the fly from conv_out, the pooled gradient and the argmax — dconv is never materialised.
#include "sbl_common.h"
#include "bf16_split.h"
extern int g_sbl_prec;      // gemm.hip: 0 = fp32 MFMA, 6 / 3 / 1 = split-bf16 products (sbl_set_matmul_precision)

#define ST_TH 8
#define ST_TW 16
#define ST_PT 5
#define ST_PH (2 * ST_TH + 5)   // 21
#define ST_PW (2 * ST_TW + 5)   // 37
#define ST_PWS 40               // padded patch row stride (floats)
#define ST_PFS (ST_PH * ST_PWS) // 840: patch frame stride
#define ST_K 245
#define ST_KP 246               // K padded to a multiple of 2 (zero weight row)
#define ST_WS 65                // weight row stride in LDS (floats): conflict-free transposed fill
#define ST_DS 68                // dconv tile row stride in LDS

__host__ __device__ constexpr int st_koff(int k) {
    // LDS patch offset of tap k = (kt*7 + kh)*7 + kw; the pad tap 245 aliases tap 244 (its weight is 0)
    return ((k < ST_K ? k : ST_K - 1) / 49) * ST_PFS + (((k < ST_K ? k : ST_K - 1) % 49) / 7) * ST_PWS +
           ((k < ST_K ? k : ST_K - 1) % 7);
}
__host__ __device__ constexpr int st_pixoff(int pix) { return (pix >> 4) * 2 * ST_PWS + (pix & 15) * 2; }

// Stage the zero-padded input patch of tile (img = n*T + t, ty, tx) into LDS.
__device__ __forceinline__ void st_load_patch(float* patch, const float* __restrict__ x, int n, int t, int ty, int tx,
                                              int T, int H, int W, int tid) {
    const int ih0 = 2 * ty * ST_TH - 3, iw0 = 2 * tx * ST_TW - 3;
    for (int i = tid; i < ST_PT * ST_PH * ST_PWS; i += 256) {
        const int c = i % ST_PWS;
        const int r = (i / ST_PWS) % ST_PH;
        const int f = i / ST_PFS;
        const int tt = t + f - 2, ih = ih0 + r, iw = iw0 + c;
        float v = 0.f;
        if (c < ST_PW && (unsigned)tt < (unsigned)T && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
            v = x[(((long)n * T + tt) * H + ih) * W + iw];
        patch[i] = v;
    }
}

// ------------------------------------------------------------------ pass 1: conv + statistics
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ out, double* __restrict__ stats, int N,
                                                            int T, int H, int W, int Ho, int Wo, int TY, int TX,
                                                            int ntiles, unsigned long long* stamp) {
    sbl_stamp_begin(stamp);
    __shared__ float Ws[ST_KP * ST_WS];
    __shared__ float patch[ST_PT * ST_PFS];
    float (*red)[128] = reinterpret_cast<float (*)[128]>(patch);   // reused after the tile loop: 80.8 KB total => 2 WG/CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;

    for (int i = tid; i < 64 * ST_K; i += 256) {   // w[co][k] -> Ws[k][co]
        const int co = i / ST_K, k = i - co * ST_K;
        Ws[k * ST_WS + co] = w[i];
    }
    if (tid < 64) Ws[ST_K * ST_WS + tid] = 0.f;

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;      // this lane's A-operand pixel inside the tile
    const int abase = py * 2 * ST_PWS + px * 2;
    const float* wsb = Ws + half * ST_WS + l31;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % TX;
        const int ty = (tile / TX) % TY;
        const int img = tile / (TX * TY);
        const int n = img / T, t = img - n * T;
        __syncthreads();   // previous tile's patch reads done (also orders the Ws fill on the first trip)
        st_load_patch(patch, x, n, t, ty, tx, T, H, W, tid);
        __syncthreads();

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < ST_KP / 2; ++ks) {
            const int off = half ? st_koff(2 * ks + 1) : st_koff(2 * ks);
            const float a = patch[abase + off];
            const float b0 = wsb[ks * 2 * ST_WS];
            const float b1 = wsb[ks * 2 * ST_WS + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
        // D: col (lane&31) = channel, row = pixel (r&3) + 8*(r>>2) + 4*half of this wave's 32 pixels
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int oh = ty * ST_TH + 2 * wave + (m >> 4), ow = tx * ST_TW + (m & 15);
            if (oh < Ho && ow < Wo) {
                float* o = out + (((long)img * Ho + oh) * Wo + ow) * 64 + l31;
                o[0] = acc0[r];
                o[32] = acc1[r];
                s1[0] += acc0[r];
                s2[0] += acc0[r] * acc0[r];
                s1[1] += acc1[r];
                s2[1] += acc1[r] * acc1[r];
            }
        }
    }
    // block reduction of the statistics, then 128 double atomics per workgroup
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s1[j] += __shfl_xor(s1[j], 32, 64);
        s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    __syncthreads();
    if (half == 0) {
        red[wave][l31] = s1[0];
        red[wave][32 + l31] = s1[1];
        red[wave][64 + l31] = s2[0];
        red[wave][96 + l31] = s2[1];
    }
    __syncthreads();
    if (tid < 128) atomicAdd(stats + tid, (double)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]));
    sbl_stamp_end(stamp);
}

// ------------------------------------------------------------------ pass 1 on the bf16 MFMA pipe (split-bf16 modes)
// Same tile, patch and epilogue; the contraction runs on v_mfma_f32_32x32x16_bf16 with both operands split exactly into bf16
// planes (bf16_split.h): 6 / 3 / 1 plane products per 32x32x16 block instead of 8 fp32 MFMAs of twice the cycles.
//   K order: k-step j = the tap rows r = 2j and 2j+1 (r = kt*7 + kh, 35 rows), 7 kw + one zero column each, so the MFMA A
//     operand of a lane (pixel l&31, k = 8*(l>>5)..+7) is EIGHT CONSECUTIVE floats of one patch row: four aligned 8-byte LDS
//     reads, split in registers - the pixel operand never takes a second trip through LDS.  18 k-steps (K = 288, 245 live).
//   Weights: split once per workgroup into LDS in B-fragment order ([k-step][channel half][plane][lane] x 16 bytes,
//     108 KB for three planes), resident for the workgroup's lifetime: one ds_read_b128 per fragment, conflict free.
//   One persistent workgroup per CU (108 KB + two 16.8 KB patch buffers); the next tile's patch is fetched into registers
//     before the contraction and written to the idle buffer after it, so the global round trip hides under 216 MFMAs.
#define SB_KS 18
__host__ __device__ constexpr int sb_rowoff(int r) { return ((r < 35 ? r : 34) / 7) * ST_PFS + ((r < 35 ? r : 34) % 7) * ST_PWS; }
#define SB_PATCH (ST_PT * ST_PFS)                  // floats
#define SB_PRE ((SB_PATCH + 255) / 256)            // patch floats per thread

// (measured and not kept: lane = patch column, wave = row phase, which makes the (frame, row) arithmetic scalar - 27 narrower
// loads per thread instead of 17 cost more than the index arithmetic saves: forward 629 -> 682 us)
__device__ __forceinline__ void sb_fetch_patch(float (&pre)[SB_PRE], const float* __restrict__ x, int tile, int TX, int TY, int T,
                                               int H, int W, int tid) {
    const int tx = tile % TX, ty = (tile / TX) % TY, img = tile / (TX * TY);
    const int n = img / T, t = img - n * T;
    const int ih0 = 2 * ty * ST_TH - 3, iw0 = 2 * tx * ST_TW - 3;
#pragma unroll
    for (int u = 0; u < SB_PRE; ++u) {
        const int i = tid + u * 256;
        const int c = i % ST_PWS, r = (i / ST_PWS) % ST_PH, f = i / ST_PFS;
        const int tt = t + f - 2, ih = ih0 + r, iw = iw0 + c;
        float v = 0.f;
        if (i < SB_PATCH && c < ST_PW && (unsigned)tt < (unsigned)T && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
            v = x[(((long)n * T + tt) * H + ih) * W + iw];
        pre[u] = v;
    }
}
__device__ __forceinline__ void sb_put_patch(float* patch, const float (&pre)[SB_PRE], int tid) {
#pragma unroll
    for (int u = 0; u < SB_PRE; ++u)
        if (tid + u * 256 < SB_PATCH) patch[tid + u * 256] = pre[u];
}

// One tile of the contraction + epilogue for one wavefront (32 pixels x 64 channels): shared by the 4-wavefront kernel and the
// 8-wavefront one below.
template <int NT>
__device__ __forceinline__ void sb_tile_compute(const float* patch, const unsigned char* wsm, int abase, int lane, int half, int l31,
                                                int wave, int ty, int tx, int img, int Ho, int Wo, float* __restrict__ out,
                                                float (&s1)[2], float (&s2)[2]) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    // software pipeline over the 18 k-steps (one wave per SIMD: nothing else hides an LDS round trip): during the MFMAs of
    // step j the wave splits the pixel floats of step j+1 (read during step j-1) and reads the weight fragments of step
    // j+1 and the pixel floats of step j+2; scheduling groups spread that work through the MFMA shadows.
    float2 raw[2][4];
    bf16x8 a[2][NPL], b[2][2][NPL];
    auto read_raw = [&](int j, float2 (&q)[4]) {
        const float2* ap = reinterpret_cast<const float2*>(patch + abase + (half ? sb_rowoff(2 * j + 1) : sb_rowoff(2 * j)));
        q[0] = ap[0]; q[1] = ap[1]; q[2] = ap[2]; q[3] = ap[3];
    };
    auto split_raw = [&](const float2 (&q)[4], bf16x8 (&f)[NPL]) {
        uint2 lo[NPL], hi[NPL];
        bf_split4<NPL>(make_float4(q[0].x, q[0].y, q[1].x, q[1].y), lo);
        bf_split4<NPL>(make_float4(q[2].x, q[2].y, q[3].x, q[3].y), hi);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f[pl] = __builtin_bit_cast(bf16x8, make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y));
    };
    auto read_b = [&](int j, bf16x8 (&f)[2][NPL]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) f[c][pl] = *reinterpret_cast<const bf16x8*>(wsm + (((j * 2 + c) * NPL + pl) * 64 + lane) * 16);
    };
    read_raw(0, raw[0]);
    read_b(0, b[0]);
    read_raw(1, raw[1]);
    split_raw(raw[0], a[0]);
#pragma unroll
    for (int j = 0; j < SB_KS; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        if (j + 1 < SB_KS) {
            read_b(j + 1, b[nxt]);
            split_raw(raw[nxt], a[nxt]);
        }
        if (j + 2 < SB_KS) read_raw(j + 2, raw[cur]);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < Tm::N; ++t) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][Tm::pa(t)], b[cur][c][Tm::pb(t)], acc[c], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2 * Tm::N; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (12 * NPL + 6 + 2 * Tm::N - 1) / (2 * Tm::N), 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    // D: col (lane&31) = channel, row = pixel (r&3) + 8*(r>>2) + 4*half of this wave's 32 pixels
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oh = ty * ST_TH + 2 * wave + (m >> 4), ow = tx * ST_TW + (m & 15);
        if (oh < Ho && ow < Wo) {
            float* o = out + (((long)img * Ho + oh) * Wo + ow) * 64 + l31;
            o[0] = acc[0][r];
            o[32] = acc[1][r];
            s1[0] += acc[0][r];
            s2[0] += acc[0][r] * acc[0][r];
            s1[1] += acc[1][r];
            s2[1] += acc[1][r] * acc[1][r];
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void stem_conv_fwd_bf_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ out, double* __restrict__ stats, int N,
                                                               int T, int H, int W, int Ho, int Wo, int TY, int TX,
                                                               int ntiles, unsigned long long* stamp) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    sbl_stamp_begin(stamp);
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];
    unsigned char* wsm = sb_smem;                                              // SB_KS * 2 * NPL fragments of 1 KB
    float* patch0 = reinterpret_cast<float*>(sb_smem + SB_KS * 2 * NPL * 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;

    // weights -> bf16 planes in fragment order: lane ln of fragment (j, c) holds channel c*32 + (ln&31), taps (row 2j + (ln>>5), kw 0..7)
    for (int u = tid; u < SB_KS * 2 * 64; u += 256) {
        const int ln = u & 63, c = (u >> 6) & 1, j = u >> 7;
        const int co = c * 32 + (ln & 31), r = 2 * j + (ln >> 5);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (r < 35 && i < 7) ? w[co * ST_K + r * 7 + i] : 0.f;
        uint2 lo[NPL], hi[NPL];
        bf_split4<NPL>(make_float4(v[0], v[1], v[2], v[3]), lo);
        bf_split4<NPL>(make_float4(v[4], v[5], v[6], v[7]), hi);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            *reinterpret_cast<uint4*>(wsm + (((j * 2 + c) * NPL + pl) * 64 + ln) * 16) = make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y);
    }

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;      // this lane's A-operand pixel inside the tile
    const int abase = py * 2 * ST_PWS + px * 2;
    float pre[SB_PRE];
    int cur = 0;
    if ((int)blockIdx.x < ntiles) {
        sb_fetch_patch(pre, x, blockIdx.x, TX, TY, T, H, W, tid);
        sb_put_patch(patch0, pre, tid);
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % TX;
        const int ty = (tile / TX) % TY;
        const int img = tile / (TX * TY);
        const float* patch = patch0 + cur * SB_PATCH;
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) sb_fetch_patch(pre, x, next, TX, TY, T, H, W, tid);

        sb_tile_compute<NT>(patch, wsm, abase, lane, half, l31, wave, ty, tx, img, Ho, Wo, out, s1, s2);
        if (next < ntiles) sb_put_patch(patch0 + (cur ^ 1) * SB_PATCH, pre, tid);
        __syncthreads();       // next patch complete; every wave is done reading this one
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s1[j] += __shfl_xor(s1[j], 32, 64);
        s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    float (*red)[128] = reinterpret_cast<float (*)[128]>(patch0);      // the loop's last barrier retired every patch read
    if (half == 0) {
        red[wave][l31] = s1[0];
        red[wave][32 + l31] = s1[1];
        red[wave][64 + l31] = s2[0];
        red[wave][96 + l31] = s2[1];
    }
    __syncthreads();
    if (tid < 128) atomicAdd(stats + tid, (double)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]));
    sbl_stamp_end(stamp);
}
template <int NT>
static int stem_launch_fwd_bf(const float* x, const float* w, float* conv_out, double* stats, int N, int T, int H, int W, int Ho,
                              int Wo, int TY, int TX, int ntiles, hipStream_t s) {
    constexpr int lds = SB_KS * 2 * BfTerms<NT>::NPL * 1024 + 2 * SB_PATCH * 4;
    static bool set[64] = {false};
    int dev = 0;
    SBL_HIP(hipGetDevice(&dev));
    if (!set[dev & 63]) {
        SBL_HIP(hipFuncSetAttribute((const void*)stem_conv_fwd_bf_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        set[dev & 63] = true;
    }
    const int grid = ntiles < 256 ? ntiles : 256;          // persistent: one workgroup per CU
    hipLaunchKernelGGL(stem_conv_fwd_bf_kernel<NT>, dim3(grid), dim3(256), lds, s, x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX,
                       ntiles, sbl_next_stamp_slot(SBL_KID_STEM));
    return 0;
}

// The same with EIGHT wavefronts on one copy of the weight planes: two groups of four wavefronts work on two tiles at a time
// (each group with its own patch buffer), so every SIMD has two wavefronts whose conversion / LDS work and MFMAs interleave.
// One patch buffer per group (two would not fit beside the 108 KB of weights): the next patch waits in registers and is
// written between two workgroup barriers after both groups are done with the current one.
template <int NT>
__global__ __launch_bounds__(512) void stem_conv_fwd_bf2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ out, double* __restrict__ stats, int N,
                                                                int T, int H, int W, int Ho, int Wo, int TY, int TX,
                                                                int ntiles, unsigned long long* stamp) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    sbl_stamp_begin(stamp);
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];
    unsigned char* wsm = sb_smem;                                              // SB_KS * 2 * NPL fragments of 1 KB
    const int tid = threadIdx.x, grp = tid >> 8, gt = tid & 255, lane = tid & 63, wave = gt >> 6, half = lane >> 5, l31 = lane & 31;
    float* patch = reinterpret_cast<float*>(sb_smem + SB_KS * 2 * NPL * 1024) + grp * SB_PATCH;
    for (int u = tid; u < SB_KS * 2 * 64; u += 512) {      // weights -> bf16 planes in fragment order (as above)
        const int ln = u & 63, c = (u >> 6) & 1, j = u >> 7;
        const int co = c * 32 + (ln & 31), r = 2 * j + (ln >> 5);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (r < 35 && i < 7) ? w[co * ST_K + r * 7 + i] : 0.f;
        uint2 lo[NPL], hi[NPL];
        bf_split4<NPL>(make_float4(v[0], v[1], v[2], v[3]), lo);
        bf_split4<NPL>(make_float4(v[4], v[5], v[6], v[7]), hi);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            *reinterpret_cast<uint4*>(wsm + (((j * 2 + c) * NPL + pl) * 64 + ln) * 16) = make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y);
    }
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;      // this lane's A-operand pixel inside the tile
    const int abase = py * 2 * ST_PWS + px * 2;
    float pre[SB_PRE];
    const int npairs = (ntiles + 1) >> 1;      // pair p = tiles 2p (group 0) and 2p + 1 (group 1); trip counts are workgroup-uniform
    if ((int)blockIdx.x < npairs && (int)blockIdx.x * 2 + grp < ntiles) {
        sb_fetch_patch(pre, x, blockIdx.x * 2 + grp, TX, TY, T, H, W, gt);
        sb_put_patch(patch, pre, gt);
    }
    __syncthreads();
    for (int pt = blockIdx.x; pt < npairs; pt += gridDim.x) {
        const int tile = pt * 2 + grp;
        const int npt = pt + (int)gridDim.x;
        const bool nvalid = npt < npairs && npt * 2 + grp < ntiles;
        if (nvalid) sb_fetch_patch(pre, x, npt * 2 + grp, TX, TY, T, H, W, gt);
        if (tile < ntiles) {      // (uniform per group of four wavefronts)
            const int tx = tile % TX;
            const int ty = (tile / TX) % TY;
            const int img = tile / (TX * TY);
            sb_tile_compute<NT>(patch, wsm, abase, lane, half, l31, wave, ty, tx, img, Ho, Wo, out, s1, s2);
        }
        __syncthreads();       // both groups are done reading their patches
        if (nvalid) sb_put_patch(patch, pre, gt);
        __syncthreads();       // next patches complete
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s1[j] += __shfl_xor(s1[j], 32, 64);
        s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    float (*red)[128] = reinterpret_cast<float (*)[128]>(sb_smem + SB_KS * 2 * NPL * 1024);      // the loop's last barrier retired every patch read
    if (half == 0) {
        red[tid >> 6][l31] = s1[0];
        red[tid >> 6][32 + l31] = s1[1];
        red[tid >> 6][64 + l31] = s2[0];
        red[tid >> 6][96 + l31] = s2[1];
    }
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tid];
        atomicAdd(stats + tid, (double)t);
    }
    sbl_stamp_end(stamp);
}
int g_sbl_stem_fwd8 = 1;      // sbl_set_tuning knob 14: 1 = the 8-wavefront forward kernel, 0 = the 4-wavefront one
template <int NT>
static int stem_launch_fwd_bf2(const float* x, const float* w, float* conv_out, double* stats, int N, int T, int H, int W, int Ho,
                               int Wo, int TY, int TX, int ntiles, hipStream_t s) {
    constexpr int lds = SB_KS * 2 * BfTerms<NT>::NPL * 1024 + 2 * SB_PATCH * 4;
    static bool set[64] = {false};
    int dev = 0;
    SBL_HIP(hipGetDevice(&dev));
    if (!set[dev & 63]) {
        SBL_HIP(hipFuncSetAttribute((const void*)stem_conv_fwd_bf2_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        set[dev & 63] = true;
    }
    const int npairs = (ntiles + 1) / 2;
    const int grid = npairs < 256 ? npairs : 256;          // persistent: one workgroup per CU
    hipLaunchKernelGGL(stem_conv_fwd_bf2_kernel<NT>, dim3(grid), dim3(512), lds, s, x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX,
                       ntiles, sbl_next_stamp_slot(SBL_KID_STEM));
    return 0;
}

// ------------------------------------------------------------------ BN finalize (shared with the trunk)
__global__ void bn_finalize_kernel(const double* __restrict__ stats, long count, float* running_mean,
                                   float* running_var, float momentum, float eps, float* save_mean, float* save_invstd,
                                   int C, long long* num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;      // nn.BatchNorm's counter, no launch of its own
    const double mean = stats[c] / (double)count;
    double var = stats[C + c] / (double)count - mean * mean;
    if (var < 0) var = 0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* mean,
                                     float* invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rm[c];
    invstd[c] = 1.0f / sqrtf(rv[c] + eps);
}

// ------------------------------------------------------------------ pass 2: BN + ReLU + 3x3/2 max-pool
// one thread = one pooled pixel x 4 channels; argmax = first maximum in (kh,kw) scan order with strict '>'
// (torch's CPU max_pool3d tie-break), stored as kh*3+kw.
__global__ __launch_bounds__(256) void stem_bn_relu_pool_kernel(const float* __restrict__ conv, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ pooled,
                                                                uint8_t* __restrict__ argmax, int NT, int Ho, int Wo, int Hp,
                                                                int Wp, unsigned long long* stamp) {
    sbl_stamp_begin(stamp);
    const long total = (long)NT * Hp * Wp * 16;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i & 15) * 4;
        long p = i >> 4;
        const int pw = p % Wp;
        p /= Wp;
        const int ph = p % Hp;
        const long img = p / Hp;
        const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c4);
        const float4 be = *reinterpret_cast<const float4*>(beta + c4);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bidx[4] = {0, 0, 0, 0};
        // branch-free window: the nine 16-byte loads are issued together (clamped addresses; taps that fall into max_pool3d's
        // padding are masked afterwards).  With a `continue` per out-of-range tap the loads sat in divergent branches and
        // were issued one by one.
        float4 v[9];
        bool ok[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = 2 * ph - 1 + kh;
            const int ihc = min(max(ih, 0), Ho - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = 2 * pw - 1 + kw;
                const int iwc = min(max(iw, 0), Wo - 1);
                ok[kh * 3 + kw] = (unsigned)ih < (unsigned)Ho && (unsigned)iw < (unsigned)Wo;
                v[kh * 3 + kw] = *reinterpret_cast<const float4*>(conv + ((img * Ho + ihc) * Wo + iwc) * 64 + c4);
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float y[4];
            y[0] = fmaxf((v[t].x - mu.x) * is.x * ga.x + be.x, 0.f);
            y[1] = fmaxf((v[t].y - mu.y) * is.y * ga.y + be.y, 0.f);
            y[2] = fmaxf((v[t].z - mu.z) * is.z * ga.z + be.z, 0.f);
            y[3] = fmaxf((v[t].w - mu.w) * is.w * ga.w + be.w, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[t] && y[q] > best[q]) {
                    best[q] = y[q];
                    bidx[q] = t;
                }
        }
        *reinterpret_cast<float4*>(pooled + (i >> 4) * 64 + c4) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<uint32_t*>(argmax + (i >> 4) * 64 + c4) =
            (uint32_t)bidx[0] | ((uint32_t)bidx[1] << 8) | ((uint32_t)bidx[2] << 16) | ((uint32_t)bidx[3] << 24);
    }
    sbl_stamp_end(stamp);
}

// gradient w.r.t. the BN output at conv pixel (oh,ow), 4 channels: max-pool adjoint (gather over the <=4 windows
// that contain the pixel, routed by the recorded argmax) times the ReLU mask (y > 0).
__device__ __forceinline__ void stem_gather_g(const float* __restrict__ dpool, const uint8_t* __restrict__ argmax, long img,
                                              int oh, int ow, int Hp, int Wp, int c4, const float y[4], float g[4]) {
    g[0] = g[1] = g[2] = g[3] = 0.f;
    // the pixel lies in the pooling windows of rows oh >> 1 and, for odd oh, (oh >> 1) + 1 (columns alike): four candidates
    // with clamped addresses and validity masks, so that the eight loads are issued together instead of inside loops whose
    // trip counts depend on the pixel
    const int ph0 = oh >> 1, pw0 = ow >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ph = min(ph0 + a, Hp - 1), pw = min(pw0 + b, Wp - 1);
            const bool valid = (a == 0 || ((oh & 1) && ph0 + 1 < Hp)) && (b == 0 || ((ow & 1) && pw0 + 1 < Wp));
            const int pos = (oh - 2 * ph + 1) * 3 + (ow - 2 * pw + 1);
            const long o = ((img * Hp + ph) * Wp + pw) * 64 + c4;
            const uint32_t am = *reinterpret_cast<const uint32_t*>(argmax + o);
            const float4 d = *reinterpret_cast<const float4*>(dpool + o);
            if (valid && (int)(am & 255) == pos) g[0] += d.x;
            if (valid && (int)((am >> 8) & 255) == pos) g[1] += d.y;
            if (valid && (int)((am >> 16) & 255) == pos) g[2] += d.z;
            if (valid && (int)(am >> 24) == pos) g[3] += d.w;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (!(y[q] > 0.f)) g[q] = 0.f;
}

// The same in two halves, so that a caller can issue the loads of several pixels before the first use: the four pool
// candidates' arg-max bytes and pooled gradients (clamped addresses), then the routing + ReLU mask.
struct StemGather {
    uint32_t am[4];
    float4 d[4];
};
__device__ __forceinline__ void stem_gather_load(const float* __restrict__ dpool, const uint8_t* __restrict__ argmax, long img, int oh,
                                                 int ow, int Hp, int Wp, int c4, StemGather& s) {
    const int ph0 = oh >> 1, pw0 = ow >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long o = ((img * Hp + min(ph0 + a, Hp - 1)) * Wp + min(pw0 + b, Wp - 1)) * 64 + c4;
            s.am[a * 2 + b] = *reinterpret_cast<const uint32_t*>(argmax + o);
            s.d[a * 2 + b] = *reinterpret_cast<const float4*>(dpool + o);
        }
}
__device__ __forceinline__ void stem_gather_apply(const StemGather& s, int oh, int ow, int Hp, int Wp, const float y[4], float g[4]) {
    g[0] = g[1] = g[2] = g[3] = 0.f;
    const int ph0 = oh >> 1, pw0 = ow >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ph = min(ph0 + a, Hp - 1), pw = min(pw0 + b, Wp - 1);
            const bool valid = (a == 0 || ((oh & 1) && ph0 + 1 < Hp)) && (b == 0 || ((ow & 1) && pw0 + 1 < Wp));
            const int pos = (oh - 2 * ph + 1) * 3 + (ow - 2 * pw + 1);
            const uint32_t am = s.am[a * 2 + b];
            const float4 d = s.d[a * 2 + b];
            if (valid && (int)(am & 255) == pos) g[0] += d.x;
            if (valid && (int)((am >> 8) & 255) == pos) g[1] += d.y;
            if (valid && (int)((am >> 16) & 255) == pos) g[2] += d.z;
            if (valid && (int)(am >> 24) == pos) g[3] += d.w;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (!(y[q] > 0.f)) g[q] = 0.f;
}

// ------------------------------------------------------------------ backward pass 1: sum g, sum g*xhat
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(const float* __restrict__ conv, const float* __restrict__ dpool,
                                                              const uint8_t* __restrict__ argmax, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, double* __restrict__ sums, int NT,
                                                              int Ho, int Wo, int Hp, int Wp, unsigned long long* stamp) {
    sbl_stamp_begin(stamp);
    __shared__ float red[16][16][8];
    const int c4 = (threadIdx.x & 15) * 4;
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c4);
    const float4 be = *reinterpret_cast<const float4*>(beta + c4);
    float sg[4] = {0, 0, 0, 0}, sgx[4] = {0, 0, 0, 0};
    // a workgroup walks whole pixel rows (img, oh): one (scalar) division per row, 16 pixels x 16 channel quads per trip and
    // SR_ROWS x SR_TRIPS trips in flight: all their loads (conv value + four pool candidates each) go out before the first
    // use (one trip at a time: 249 us, three: 170 us; the traffic is 604 MB = 96 us)
    constexpr int SR_TRIPS = 3, SR_ROWS = 1;      // (44-pixel rows: three trips of 16 pixels; two rows at a time need 197 registers: 187 us)
    const int nrows = NT * Ho;
    for (int r0 = blockIdx.x * SR_ROWS; r0 < nrows; r0 += gridDim.x * SR_ROWS) {
        for (int ow0 = 0; ow0 < Wo; ow0 += 16 * SR_TRIPS) {
            float4 v[SR_ROWS][SR_TRIPS];
            StemGather sg_[SR_ROWS][SR_TRIPS];
#pragma unroll
            for (int j = 0; j < SR_ROWS; ++j) {
                const int r = min(r0 + j, nrows - 1);
                const int img = r / Ho, oh = r - img * Ho;
#pragma unroll
                for (int u = 0; u < SR_TRIPS; ++u) {
                    const int ow = min(ow0 + 16 * u + (int)(threadIdx.x >> 4), Wo - 1);
                    v[j][u] = *reinterpret_cast<const float4*>(conv + ((long)r * Wo + ow) * 64 + c4);
                    stem_gather_load(dpool, argmax, img, oh, ow, Hp, Wp, c4, sg_[j][u]);
                }
            }
#pragma unroll
            for (int j = 0; j < SR_ROWS; ++j) {
                const int r = r0 + j;
                const int img = r / Ho, oh = r - img * Ho;
#pragma unroll
                for (int u = 0; u < SR_TRIPS; ++u) {
                    const int ow = ow0 + 16 * u + (int)(threadIdx.x >> 4);
                    if (r < nrows && ow < Wo) {
                        const float4 vv = v[j][u];
                        float xh[4] = {(vv.x - mu.x) * is.x, (vv.y - mu.y) * is.y, (vv.z - mu.z) * is.z, (vv.w - mu.w) * is.w};
                        float y[4] = {xh[0] * ga.x + be.x, xh[1] * ga.y + be.y, xh[2] * ga.z + be.z, xh[3] * ga.w + be.w};
                        float g[4];
                        stem_gather_apply(sg_[j][u], oh, ow, Hp, Wp, y, g);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            sg[k] += g[k];
                            sgx[k] += g[k] * xh[k];
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x >> 4][threadIdx.x & 15][k] = sg[k];
        red[threadIdx.x >> 4][threadIdx.x & 15][4 + k] = sgx[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cq = threadIdx.x >> 3, k = threadIdx.x & 7;   // channel-quad, component
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][cq][k];
        const int ch = cq * 4 + (k & 3);
        atomicAdd(sums + (k < 4 ? ch : 64 + ch), (double)s);
    }
    sbl_stamp_end(stamp);
}

// ------------------------------------------------------------------ backward pass 2: weight gradient
// dw[co][k] = sum_pixels dconv[pixel][co] * patch(pixel, k),  dconv = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).
// MFMA: rows = co (2 tiles), cols = k (8 tiles of 32, 245 valid), contraction over the tile's 128 pixels.
// Wave w owns k-tiles {2w, 2w+1}; accumulators persist across the workgroup's tiles, float atomics at the end.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ conv,
                                                         const float* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const double* __restrict__ sums, float* __restrict__ dw,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int T,
                                                         int H, int W, int Ho, int Wo, int Hp, int Wp, int TY, int TX,
                                                         int ntiles, unsigned long long* stamp) {
    sbl_stamp_begin(stamp);
    __shared__ __attribute__((aligned(16))) float Ds[128 * ST_DS];
    __shared__ float patch[ST_PT * ST_PFS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const double cnt = (double)N * T * Ho * Wo;
    if (blockIdx.x == 0 && tid < 64) {
        dbeta[tid] = (float)sums[tid];
        dgamma[tid] = (float)sums[64 + tid];
    }
    const int c4 = (tid & 15) * 4;
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c4);
    const float4 be = *reinterpret_cast<const float4*>(beta + c4);
    float mg[4], mgx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mg[k] = (float)(sums[c4 + k] / cnt);
        mgx[k] = (float)(sums[64 + c4 + k] / cnt);
    }
    const int k0 = (2 * wave) * 32 + l31, k1 = k0 + 32;   // this lane's two B-operand taps
    const int koff0 = st_koff(k0), koff1 = st_koff(k1);

    f32x16 acc[2][2];   // [co tile][k tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % TX;
        const int ty = (tile / TX) % TY;
        const int img = tile / (TX * TY);
        const int n = img / T, t = img - n * T;
        __syncthreads();
        st_load_patch(patch, x, n, t, ty, tx, T, H, W, tid);
        // dconv tile -> Ds[pixel][co]; thread = (pixel group, channel quad), 8 passes of 16 pixels
#pragma unroll 2
        for (int ps = 0; ps < 8; ++ps) {
            const int pix = ps * 16 + (tid >> 4);
            const int oh = ty * ST_TH + (pix >> 4), ow = tx * ST_TW + (pix & 15);
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oh < Ho && ow < Wo) {
                const float4 v = *reinterpret_cast<const float4*>(conv + (((long)img * Ho + oh) * Wo + ow) * 64 + c4);
                float xh[4] = {(v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w};
                float y[4] = {xh[0] * ga.x + be.x, xh[1] * ga.y + be.y, xh[2] * ga.z + be.z, xh[3] * ga.w + be.w};
                float g[4];
                stem_gather_g(dpool, argmax, img, oh, ow, Hp, Wp, c4, y, g);
                d.x = ga.x * is.x * (g[0] - mg[0] - xh[0] * mgx[0]);
                d.y = ga.y * is.y * (g[1] - mg[1] - xh[1] * mgx[1]);
                d.z = ga.z * is.z * (g[2] - mg[2] - xh[2] * mgx[2]);
                d.w = ga.w * is.w * (g[3] - mg[3] - xh[3] * mgx[3]);
            }
            *reinterpret_cast<float4*>(&Ds[pix * ST_DS + c4]) = d;
        }
        __syncthreads();
#pragma unroll 8
        for (int ps = 0; ps < 64; ++ps) {
            const int pix = 2 * ps + half;
            const int poff = st_pixoff(pix);
            const float a0 = Ds[pix * ST_DS + l31];
            const float a1 = Ds[pix * ST_DS + 32 + l31];
            const float b0 = patch[poff + koff0];
            const float b1 = patch[poff + koff1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // D: col (lane&31) = tap within the k tile, row = co within the co tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (2 * wave + j) * 32 + l31;
            if (k < ST_K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    atomicAdd(dw + co * ST_K + k, acc[i][j][r]);
                }
            }
        }
    sbl_stamp_end(stamp);
}

// ------------------------------------------------------------------ backward pass 2 on the bf16 MFMA pipe (split-bf16 modes)
// Same tiles, dconv recomputation and atomics; the contraction over the tile's 128 pixels runs as 8 k-steps (one 16-pixel row
// each) of v_mfma_f32_32x32x16_bf16.  dconv sits TRANSPOSED in LDS ([co][pixel], rows of 132 floats) so a lane's A operand
// (channel l&31, pixels 8*(l>>5)..+7 of the row) is two aligned 16-byte reads; the B operand (tap l&31 of the wave's two
// 32-tap tiles, the same 8 pixels) is eight stride-2 floats of the patch.  Both are split into bf16 planes in registers.
#define SB_DT 132
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void stem_wgrad_bf_kernel(const float* __restrict__ x, const float* __restrict__ conv,
                                                            const float* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const double* __restrict__ sums, float* __restrict__ dw,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int T,
                                                            int H, int W, int Ho, int Wo, int Hp, int Wp, int TY, int TX,
                                                            int ntiles, unsigned long long* stamp) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    sbl_stamp_begin(stamp);
    __shared__ __attribute__((aligned(16))) float Dt[64 * SB_DT];
    __shared__ float patch[ST_PT * ST_PFS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const double cnt = (double)N * T * Ho * Wo;
    if (blockIdx.x == 0 && tid < 64) {
        dbeta[tid] = (float)sums[tid];
        dgamma[tid] = (float)sums[64 + tid];
    }
    const int c4 = (tid & 15) * 4;
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c4);
    const float4 be = *reinterpret_cast<const float4*>(beta + c4);
    float mg[4], mgx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mg[k] = (float)(sums[c4 + k] / cnt);
        mgx[k] = (float)(sums[64 + c4 + k] / cnt);
    }
    const int k0 = (2 * wave) * 32 + l31, k1 = k0 + 32;   // this lane's two B-operand taps
    const int boff0 = st_koff(k0) + 16 * half, boff1 = st_koff(k1) + 16 * half;      // + its 8 pixels' first column
    const int aoff = l31 * SB_DT + 8 * half;

    f32x16 acc[2][2];   // [co tile][k tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % TX;
        const int ty = (tile / TX) % TY;
        const int img = tile / (TX * TY);
        const int n = img / T, t = img - n * T;
        __syncthreads();
        st_load_patch(patch, x, n, t, ty, tx, T, H, W, tid);
        // dconv tile -> Dt[co][pixel]; thread = (pixel group, channel quad), 8 passes of 16 pixels
#pragma unroll 2
        for (int ps = 0; ps < 8; ++ps) {
            const int pix = ps * 16 + (tid >> 4);
            const int oh = ty * ST_TH + (pix >> 4), ow = tx * ST_TW + (pix & 15);
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oh < Ho && ow < Wo) {
                const float4 v = *reinterpret_cast<const float4*>(conv + (((long)img * Ho + oh) * Wo + ow) * 64 + c4);
                float xh[4] = {(v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w};
                float y[4] = {xh[0] * ga.x + be.x, xh[1] * ga.y + be.y, xh[2] * ga.z + be.z, xh[3] * ga.w + be.w};
                float g[4];
                stem_gather_g(dpool, argmax, img, oh, ow, Hp, Wp, c4, y, g);
                d.x = ga.x * is.x * (g[0] - mg[0] - xh[0] * mgx[0]);
                d.y = ga.y * is.y * (g[1] - mg[1] - xh[1] * mgx[1]);
                d.z = ga.z * is.z * (g[2] - mg[2] - xh[2] * mgx[2]);
                d.w = ga.w * is.w * (g[3] - mg[3] - xh[3] * mgx[3]);
            }
            Dt[(c4 + 0) * SB_DT + pix] = d.x;
            Dt[(c4 + 1) * SB_DT + pix] = d.y;
            Dt[(c4 + 2) * SB_DT + pix] = d.z;
            Dt[(c4 + 3) * SB_DT + pix] = d.w;
        }
        __syncthreads();
#pragma unroll 2
        for (int sr = 0; sr < 8; ++sr) {          // k-step = pixel row sr of the tile
            bf16x8 a[2][NPL], b[2][NPL];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4* ap = reinterpret_cast<const float4*>(Dt + i * 32 * SB_DT + aoff + sr * 16);
                uint2 lo[NPL], hi[NPL];
                bf_split4<NPL>(ap[0], lo);
                bf_split4<NPL>(ap[1], hi);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[i][pl] = __builtin_bit_cast(bf16x8, make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* bp = patch + sr * 2 * ST_PWS + (j ? boff1 : boff0);
                uint2 lo[NPL], hi[NPL];
                bf_split4<NPL>(make_float4(bp[0], bp[2], bp[4], bp[6]), lo);
                bf_split4<NPL>(make_float4(bp[8], bp[10], bp[12], bp[14]), hi);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) b[j][pl] = __builtin_bit_cast(bf16x8, make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tt = 0; tt < Tm::N; ++tt)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][Tm::pa(tt)], b[j][Tm::pb(tt)], acc[i][j], 0, 0, 0);
        }
    }
    // D: col (lane&31) = tap within the k tile, row = co within the co tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (2 * wave + j) * 32 + l31;
            if (k < ST_K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    atomicAdd(dw + co * ST_K + k, acc[i][j][r]);
                }
            }
        }
    sbl_stamp_end(stamp);
}

// ------------------------------------------------------------------ backward pass 2, split-bf16 modes, operands split ONCE
// The kernel above splits every operand where it is used: each dconv value in all four wavefronts, each patch value once per
// tap it feeds (~60 times), 216 conversion instructions per 24 MFMAs.  Here both operands are split into bf16 planes on the
// way INTO LDS and every MFMA operand comes out of two transposed reads (ds_read_b64_tr_b16: per 16 lanes a block of
// 4 K-rows x 16 columns; each lane supplies the 8-byte-aligned address of 4 consecutive columns of one row):
//  * A = dconv: pixel-major planes [64 pixels][64 co] (half a tile at a time: 4 pixel rows = 4 k-steps), 128-byte rows with
//    the two 64-byte halves swapped on rows with bit 1 set (conflict-free for 4 consecutive rows).
//  * B = taps: the columns of a block are taps; the 7 kw taps (+ 1 pad) of a (kt, kh) pair at pixel px are the 8 CONSECUTIVE
//    patch values from column 2 px of patch row 2 py + kh, so a lane's "4 consecutive columns" are 4 consecutive patch
//    values at element 2 px + {0, 4}: 8-byte aligned for even px.  Odd pixels read a second copy of the planes shifted by two
//    elements.  A 32-column tile = 4 (kt, kh) pairs; 35 pairs -> 9 tiles (288 columns, 245 kept).
// 4 wavefronts: wavefront w owns tap tiles w and w + 4 for both co tiles and every other k-step of one co tile of the ninth
// tap tile; 27 MFMAs per 12 + 15 transposed reads at bf16x6 (average) and no conversion in the K loop.  77 KB of LDS and
// 4-wavefront workgroups: two per CU.
// MEASURED (tools/bench_stem.py, knobs 12 / 13): 642 us against 801 us for the kernel above.  Phase ablation: k-steps 240 us
// (MFMA-bound on the SIMDs of wavefronts 0 and 1: 30 MFMAs per k-step, 65 tiles per CU), dconv recompute 250 us (VALU: the
// pool / ReLU / BatchNorm adjoint of 2048 (pixel, 4 channels) items per tile), patch staging 146 us - and the three still
// ADD UP: a workgroup's phases are serial, and the second workgroup of the CU does not fill the other pipe (starting half of
// the workgroups 0.5 - 2 us late changes nothing).  A first version with 6-wavefront workgroups (tap tiles j, j + 3, j + 6 per
// wavefront) measured 794 us: two 6-wavefront workgroups do NOT co-reside on a CU even when registers and LDS allow it
// (tools/probes/coresidency.hip: 4-wavefront workgroups do; the second workgroup's wavefronts are placed from SIMD 0 again,
// 2 + 2 wavefronts of ~160 registers do not fit one SIMD), so that version ran one workgroup per CU.  Batching two recompute
// items (48 more registers) spilled at 168 registers and took 1.2 ms.
#define SW_RS 40                                   // patch row stride in elements (80 bytes: 8-byte aligned rows)
#define SW_PLANE (ST_PT * ST_PH * SW_RS * 2)       // bytes of one patch plane (8400)
#define SW_DPLANE (64 * 128)                       // bytes of one dconv plane (half tile)
#define SW_THREADS 256
#define SW_PQ 5                                      // patch quads per thread (5 * 21 * 10 = 1050 <= 5 * 256)
__device__ __forceinline__ int sw_off(int row, int f) {      // 8-byte slot of channels 4f .. 4f + 3 of dconv row `row`
    return row * 128 + ((((f >> 1) ^ (((row >> 1) & 1) << 2))) << 4) + ((f & 1) << 3);
}
template <int NT>
__global__ __launch_bounds__(SW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_wgrad_tr_kernel(const float* __restrict__ x, const float* __restrict__ conv,
                                                                  const float* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const double* __restrict__ sums, float* __restrict__ dw,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int T,
                                                                  int H, int W, int Ho, int Wo, int Hp, int Wp, int TY, int TX,
                                                                  int ntiles, unsigned long long* stamp, int ablate) {
    using Tm = BfTerms<NT>;
    constexpr int NPL = Tm::NPL;
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    sbl_stamp_begin(stamp);
    unsigned char* pe = sw_smem;                              // patch planes, copy E then copy O (shifted by two elements)
    unsigned char* dp = sw_smem + 2 * NPL * SW_PLANE;         // dconv planes of the current half tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const double cnt = (double)N * T * Ho * Wo;
    if (blockIdx.x == 0 && tid < 64) {
        dbeta[tid] = (float)sums[tid];
        dgamma[tid] = (float)sums[64 + tid];
    }
    const int c4 = (tid & 15) * 4;
    // per-channel constants of the BatchNorm adjoint in LDS:
    // [0] mean [1] invstd [2] gamma [3] beta [4] mean(g) [5] mean(g * xhat)
    float* kc = reinterpret_cast<float*>(sw_smem + 2 * NPL * SW_PLANE + NPL * SW_DPLANE);
    if (tid < 64) {
        kc[0 * 64 + tid] = mean[tid];
        kc[1 * 64 + tid] = invstd[tid];
        kc[2 * 64 + tid] = gamma[tid];
        kc[3 * 64 + tid] = beta[tid];
        kc[4 * 64 + tid] = (float)(sums[tid] / cnt);
        kc[5 * 64 + tid] = (float)(sums[64 + tid] / cnt);
    }
    // Wavefront w: tap tiles w and w + 4 for both co tiles (4 accumulators); the ninth tap tile's two blocks are shared: co
    // tile w & 1, the k-steps with (step & 1) == (w >> 1) - 27 MFMAs per k-step and wavefront on average at bf16x6.  4 wavefronts, not 6: two 6-wavefront workgroups do not
    // co-reside on a CU (tools/probes/coresidency.hip: the second workgroup's wavefronts start at SIMD 0 again).
    // A operand: rows 16 s + 8 (g >> 1) + q (+ 4) of the half tile, channels 32 i + 16 (g & 1) + 4 p
    int aoff0[2], aoff1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int fA = 8 * i + 4 * (g4 & 1) + p4;
        aoff0[i] = sw_off(8 * (g4 >> 1) + q4, fA);      // step s adds 2048 bytes
        aoff1[i] = sw_off(8 * (g4 >> 1) + q4 + 4, fA);
    }
    // B operand: pixel px = 8 (g >> 1) + q (+ 4: 16 bytes further, same copy), columns 16 (g & 1) + 4 p .. + 3 of the tile =
    // pair 4 tile + 2 (g & 1) + (p >> 1), kw = 4 (p & 1) .. + 3
    int boff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int tl = u < 2 ? wave + 4 * u : 8;
        const int pair = min(4 * tl + 2 * (g4 & 1) + (p4 >> 1), 34);      // pair 35 (pad) aliases 34: its columns are dropped
        const int kt = pair / 7, kh = pair - kt * 7;
        const int px = 8 * (g4 >> 1) + q4;
        boff[u] = (q4 & 1) * (NPL * SW_PLANE) + ((kt * ST_PH + kh) * SW_RS + 2 * px + 4 * (p4 & 1) + 2 * (q4 & 1)) * 2;
    }

    f32x16 acc[2][2], accx;      // [tap tile slot][co tile], and the ninth tile's block
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[0][0][r] = acc[0][1][r] = acc[1][0][r] = acc[1][1][r] = 0.f;
        accx[r] = 0.f;
    }

    // the tile's input patch as quads of 4 consecutive columns, all loads of a thread issued together (clamped addresses, the
    // out-of-range values replaced by zeros afterwards: no branch around a load)
    float4 pv[SW_PQ];
    auto fetch_patch = [&](int tile) {
        const int tx = tile % TX;
        const int ty = (tile / TX) % TY;
        const int img = tile / (TX * TY);
        const int n = img / T, t = img - n * T;
        const int ih0 = 2 * ty * ST_TH - 3, iw0 = 2 * tx * ST_TW - 3;
#pragma unroll
        for (int u = 0; u < SW_PQ; ++u) {
            const int i = min(tid + u * SW_THREADS, ST_PT * ST_PH * (SW_RS / 4) - 1);
            const int j = i % (SW_RS / 4);
            const int r = (i / (SW_RS / 4)) % ST_PH;
            const int f = i / (ST_PH * (SW_RS / 4));
            const int tt = t + f - 2, ih = ih0 + r, iw = iw0 + 4 * j;
            const bool rok = (unsigned)tt < (unsigned)T && (unsigned)ih < (unsigned)H;
            const float* row = x + (((long)n * T + min(max(tt, 0), T - 1)) * H + min(max(ih, 0), H - 1)) * W;
            const float a0 = row[min(max(iw + 0, 0), W - 1)], a1 = row[min(max(iw + 1, 0), W - 1)];
            const float a2 = row[min(max(iw + 2, 0), W - 1)], a3 = row[min(max(iw + 3, 0), W - 1)];
            pv[u].x = rok && (unsigned)(iw + 0) < (unsigned)W ? a0 : 0.f;
            pv[u].y = rok && (unsigned)(iw + 1) < (unsigned)W ? a1 : 0.f;
            pv[u].z = rok && (unsigned)(iw + 2) < (unsigned)W ? a2 : 0.f;
            pv[u].w = rok && (unsigned)(iw + 3) < (unsigned)W ? a3 : 0.f;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch_patch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % TX;
        const int ty = (tile / TX) % TY;
        const int img = tile / (TX * TY);
        __syncthreads();      // every wavefront is done with the previous tile's planes
        {   // patch registers (fetched under the previous tile's k-steps) -> planes; copy E at element 4 j, copy O at element 4 j + 2
#pragma unroll
            for (int u = 0; u < SW_PQ; ++u) {
                const int i = tid + u * SW_THREADS;
                if (i < ST_PT * ST_PH * (SW_RS / 4)) {
                    const int j = i % (SW_RS / 4);
                    uint2 pl[NPL];
                    bf_split4<NPL>(pv[u], pl);
                    const int eo = ((i / (SW_RS / 4)) * SW_RS + 4 * j) * 2;
#pragma unroll
                    for (int pq = 0; pq < NPL; ++pq) {
                        *reinterpret_cast<uint2*>(pe + pq * SW_PLANE + eo) = pl[pq];
                        *reinterpret_cast<unsigned*>(pe + (NPL + pq) * SW_PLANE + eo + 4) = pl[pq].x;
                        // (the last quad of a row is padding - columns 36 .. 39, only 36 is data and copy O never reads it there)
                        if (j < SW_RS / 4 - 1) *reinterpret_cast<unsigned*>(pe + (NPL + pq) * SW_PLANE + eo + 8) = pl[pq].y;
                        if (j == 0) *reinterpret_cast<unsigned*>(pe + (NPL + pq) * SW_PLANE + eo) = 0u;      // elements 0, 1 of copy O: never data, kept finite
                    }
                }
            }
        }
        if (tile + (int)gridDim.x < ntiles && !(ablate & 4)) fetch_patch(tile + gridDim.x);      // in flight under this tile's k-steps
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            if (hf) __syncthreads();      // the first half's k-steps are done with the dconv planes
            // dconv of pixel rows 4 hf .. 4 hf + 3 -> planes.  Thread = (pixel column tid >> 4, channel quad tid & 15); item u of
            // the thread is pixel row u of the half, so everything that depends on the column or the channels only is computed
            // once per half: the column's two pool-window candidates (index, validity, position inside the window), the channel
            // constants of the BatchNorm adjoint, the LDS slot.  The nine loads of an item (conv value, four pool candidates:
            // arg-max bytes + pooled gradient) are issued together.
            if (!(ablate & 8)) {
                const int pcol = tid >> 4;
                const int ow = tx * ST_TW + pcol, owc = min(ow, Wo - 1);
                const int pw0 = owc >> 1;
                const int pwb[2] = {min(pw0, Wp - 1), min(pw0 + 1, Wp - 1)};
                const bool vb[2] = {true, (owc & 1) && pw0 + 1 < Wp};
                const int posb[2] = {owc - 2 * pwb[0] + 1, owc - 2 * pwb[1] + 1};
                const float4 mu = *reinterpret_cast<const float4*>(kc + 0 * 64 + c4), is = *reinterpret_cast<const float4*>(kc + 1 * 64 + c4);
                const float4 ga = *reinterpret_cast<const float4*>(kc + 2 * 64 + c4), be = *reinterpret_cast<const float4*>(kc + 3 * 64 + c4);
                const float4 mg = *reinterpret_cast<const float4*>(kc + 4 * 64 + c4), mgx = *reinterpret_cast<const float4*>(kc + 5 * 64 + c4);
                const float4 gi = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
#pragma unroll 1
                for (int u = 0; u < 4; ++u) {
                    const int oh = ty * ST_TH + 4 * hf + u, ohc = min(oh, Ho - 1);
                    const int ph0 = ohc >> 1;
                    const int pha[2] = {min(ph0, Hp - 1), min(ph0 + 1, Hp - 1)};
                    const bool va[2] = {true, (ohc & 1) && ph0 + 1 < Hp};
                    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f), dq[4];
                    uint32_t am[4];
                    if (!(ablate & 2)) {
                        cv = *reinterpret_cast<const float4*>(conv + (((long)img * Ho + ohc) * Wo + owc) * 64 + c4);
#pragma unroll
                        for (int ca = 0; ca < 2; ++ca) {
                            const long rowo = ((long)img * Hp + pha[ca]) * Wp;
#pragma unroll
                            for (int cb2 = 0; cb2 < 2; ++cb2) {
                                const long o = (rowo + pwb[cb2]) * 64 + c4;
                                am[ca * 2 + cb2] = *reinterpret_cast<const uint32_t*>(argmax + o);
                                dq[ca * 2 + cb2] = *reinterpret_cast<const float4*>(dpool + o);
                            }
                        }
                    }
                    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (oh < Ho && ow < Wo) {
                        const float xh[4] = {(cv.x - mu.x) * is.x, (cv.y - mu.y) * is.y, (cv.z - mu.z) * is.z, (cv.w - mu.w) * is.w};
                        const float y[4] = {xh[0] * ga.x + be.x, xh[1] * ga.y + be.y, xh[2] * ga.z + be.z, xh[3] * ga.w + be.w};
                        // max-pool adjoint routed by the recorded arg-max (stem_gather_g's rule) times the ReLU mask
                        float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
                            for (int cb2 = 0; cb2 < 2; ++cb2) {
                                // position of the pixel inside the candidate window, or 15 (no arg-max byte has that value) when the
                                // candidate is not a window of this pixel
                                const int pos = (va[ca] && vb[cb2]) ? (oh - 2 * pha[ca] + 1) * 3 + posb[cb2] : 15;
                                const uint32_t m = am[ca * 2 + cb2];
                                const float4 dd = dq[ca * 2 + cb2];
                                if ((int)(m & 255) == pos) g[0] += dd.x;
                                if ((int)((m >> 8) & 255) == pos) g[1] += dd.y;
                                if ((int)((m >> 16) & 255) == pos) g[2] += dd.z;
                                if ((int)(m >> 24) == pos) g[3] += dd.w;
                            }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (!(y[k] > 0.f)) g[k] = 0.f;
                        d.x = gi.x * (g[0] - mg.x - xh[0] * mgx.x);
                        d.y = gi.y * (g[1] - mg.y - xh[1] * mgx.y);
                        d.z = gi.z * (g[2] - mg.z - xh[2] * mgx.z);
                        d.w = gi.w * (g[3] - mg.w - xh[3] * mgx.w);
                    }
                    uint2 pl[NPL];
                    bf_split4<NPL>(d, pl);
                    const int off = sw_off(16 * u + pcol, tid & 15);
#pragma unroll
                    for (int pq = 0; pq < NPL; ++pq) *reinterpret_cast<uint2*>(dp + pq * SW_DPLANE + off) = pl[pq];
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int sr = (ablate & 1) ? 4 : 0; sr < 4; ++sr) {          // k-step = pixel row 4 hf + sr of the tile
                bf16x8 a[2][NPL];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pq = 0; pq < NPL; ++pq) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dp + pq * SW_DPLANE + sr * 2048 + aoff0[i]));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dp + pq * SW_DPLANE + sr * 2048 + aoff1[i]));
                        a[i][pq] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                const int rowb = (4 * hf + sr) * 2 * SW_RS * 2;      // patch row 2 py: bytes
                auto tap_frag = [&](int u, bf16x8 (&b)[NPL]) {
#pragma unroll
                    for (int pq = 0; pq < NPL; ++pq) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(pe + pq * SW_PLANE + rowb + boff[u]));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(pe + pq * SW_PLANE + rowb + boff[u] + 16));
                        b[pq] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                };
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 b[NPL];
                    tap_frag(u, b);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int tt = 0; tt < Tm::N; ++tt)
                            acc[u][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][Tm::pa(tt)], b[Tm::pb(tt)], acc[u][i], 0, 0, 0);
                }
                if ((sr & 1) == (wave >> 1)) {      // wavefront-uniform branch: every lane of the wavefront takes it (the transposed reads need a full EXEC)
                    bf16x8 b[NPL];
                    tap_frag(2, b);
#pragma unroll
                    for (int tt = 0; tt < Tm::N; ++tt) {
                        if ((wave & 1) == 0) accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][Tm::pa(tt)], b[Tm::pb(tt)], accx, 0, 0, 0);
                        else accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][Tm::pa(tt)], b[Tm::pb(tt)], accx, 0, 0, 0);
                    }
                }
            }
        }
    }
    // D: column l & 31 = (pair slot, kw) of the tap tile, row = co within the co tile
    auto flush = [&](const f32x16& v, int tl, int i) {
        const int pair = 4 * tl + (l31 >> 3), kw = l31 & 7;
        if (pair < 35 && kw < 7) {
            const int k = pair * 7 + kw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                atomicAdd(dw + co * ST_K + k, v[r]);
            }
        }
    };
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) flush(acc[u][i], wave + 4 * u, i);
    flush(accx, 8, wave & 1);      // (two wavefronts per co tile: partial sums over alternate k-steps)
    sbl_stamp_end(stamp);
}
int g_sbl_stem_ablate = 0;        // knob 13 (measurement only, WRONG RESULTS): stem_wgrad_tr_kernel without 1 the k-steps, 2 the dconv loads, 4 the patch loads, 8 the dconv phase
int g_sbl_stem_wgrad_tr = 1;      // sbl_set_tuning knob 12: 1 = the kernel above (default), 0 = stem_wgrad_bf_kernel

// ------------------------------------------------------------------ host entry points
static int stem_dims(const char* who, int N, int T, int H, int W) {
    SBL_REQUIRE(N > 0 && T > 0 && H >= 8 && W >= 8 && H % 4 == 0 && W % 4 == 0, "%s: bad clip dims N=%d T=%d H=%d W=%d (H,W multiples of 4)", who, N, T, H, W);
    SBL_REQUIRE((long)N * T * (H / 2) * (W / 2) * 64 < (1L << 40), "%s: too large", who);
    return 0;
}

extern "C" int sbl_stem_conv_fwd(const float* x, const float* w, float* conv_out, double* stats, int N, int T, int H,
                                 int W, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = stem_dims("sbl_stem_conv_fwd", N, T, H, W)) return e;
    SBL_REQUIRE(x && w && conv_out && stats, "sbl_stem_conv_fwd: null pointer");
    const int Ho = H / 2, Wo = W / 2, TY = sbl_cdiv(Ho, ST_TH), TX = sbl_cdiv(Wo, ST_TW);
    const long ntiles = (long)N * T * TY * TX;
    SBL_REQUIRE(ntiles < (1L << 31), "sbl_stem_conv_fwd: too many tiles");
    SBL_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 128, s));
    if (g_sbl_prec) {
        int e;
        if (g_sbl_stem_fwd8)
            e = g_sbl_prec == 6 ? stem_launch_fwd_bf2<6>(x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX, (int)ntiles, s)
              : g_sbl_prec == 3 ? stem_launch_fwd_bf2<3>(x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX, (int)ntiles, s)
                                : stem_launch_fwd_bf2<1>(x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX, (int)ntiles, s);
        else
            e = g_sbl_prec == 6 ? stem_launch_fwd_bf<6>(x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX, (int)ntiles, s)
              : g_sbl_prec == 3 ? stem_launch_fwd_bf<3>(x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX, (int)ntiles, s)
                                : stem_launch_fwd_bf<1>(x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY, TX, (int)ntiles, s);
        if (e) return e;
        SBL_LAUNCH_CHECK("sbl_stem_conv_fwd(bf16)");
        return 0;
    }
    const int grid = (int)(ntiles < 512 ? ntiles : 512);   // persistent: 2 workgroups per CU (LDS-bound)
    hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(grid), dim3(256), 0, s, x, w, conv_out, stats, N, T, H, W, Ho, Wo, TY,
                       TX, (int)ntiles, sbl_next_stamp_slot(SBL_KID_STEM));
    SBL_LAUNCH_CHECK("sbl_stem_conv_fwd");
    return 0;
}

extern "C" int sbl_bn_finalize(const double* stats, long count, float* running_mean, float* running_var, float momentum,
                               float eps, float* save_mean, float* save_invstd, int C, int64_t* num_batches_tracked,
                               sbl_stream_t stream) {
    SBL_REQUIRE(stats && save_mean && save_invstd && C > 0 && count > 0, "sbl_bn_finalize: bad args");
    SBL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sbl_bn_finalize: running stats must both be set or both null");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sbl_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, stats, count,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, C, (long long*)num_batches_tracked);
    SBL_LAUNCH_CHECK("sbl_bn_finalize");
    return 0;
}
extern "C" int sbl_bn_eval_stats(const float* rm, const float* rv, float eps, float* mean, float* invstd, int C,
                                 sbl_stream_t stream) {
    SBL_REQUIRE(rm && rv && mean && invstd && C > 0, "sbl_bn_eval_stats: bad args");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(sbl_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, rm, rv, eps, mean,
                       invstd, C);
    SBL_LAUNCH_CHECK("sbl_bn_eval_stats");
    return 0;
}

extern "C" int sbl_stem_bn_relu_pool_fwd(const float* conv_out, const float* mean, const float* invstd, const float* gamma,
                                         const float* beta, float* pooled, uint8_t* argmax, int NT, int Ho, int Wo,
                                         sbl_stream_t stream) {
    SBL_REQUIRE(conv_out && mean && invstd && gamma && beta && pooled && argmax, "sbl_stem_bn_relu_pool_fwd: null pointer");
    SBL_REQUIRE(NT > 0 && Ho >= 2 && Wo >= 2 && Ho % 2 == 0 && Wo % 2 == 0, "sbl_stem_bn_relu_pool_fwd: bad dims %d %d %d", NT, Ho, Wo);
    const int Hp = Ho / 2, Wp = Wo / 2;
    const long total = (long)NT * Hp * Wp * 16;
    const int grid = (int)(sbl_cdiv(total, 256) > 8192 ? 8192 : sbl_cdiv(total, 256));
    hipLaunchKernelGGL(stem_bn_relu_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, conv_out, mean, invstd,
                       gamma, beta, pooled, argmax, NT, Ho, Wo, Hp, Wp, sbl_next_stamp_slot(SBL_KID_STEM));
    SBL_LAUNCH_CHECK("sbl_stem_bn_relu_pool_fwd");
    return 0;
}

extern "C" int sbl_stem_bwd_reduce(const float* conv_out, const float* dpooled, const uint8_t* argmax, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, double* sums, int NT, int Ho,
                                   int Wo, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(conv_out && dpooled && argmax && mean && invstd && gamma && beta && sums, "sbl_stem_bwd_reduce: null pointer");
    SBL_REQUIRE(NT > 0 && Ho >= 2 && Wo >= 2 && Ho % 2 == 0 && Wo % 2 == 0, "sbl_stem_bwd_reduce: bad dims");
    SBL_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 128, s));
    const long npix = (long)NT * Ho * Wo;
    const int grid = (int)(sbl_cdiv(npix, 16) > 4096 ? 4096 : sbl_cdiv(npix, 16));
    hipLaunchKernelGGL(stem_bwd_reduce_kernel, dim3(grid), dim3(256), 0, s, conv_out, dpooled, argmax, mean, invstd, gamma,
                       beta, sums, NT, Ho, Wo, Ho / 2, Wo / 2, sbl_next_stamp_slot(SBL_KID_STEM));
    SBL_LAUNCH_CHECK("sbl_stem_bwd_reduce");
    return 0;
}

extern "C" int sbl_stem_wgrad(const float* x, const float* conv_out, const float* dpooled, const uint8_t* argmax,
                              const float* mean, const float* invstd, const float* gamma, const float* beta,
                              const double* sums, float* dw, float* dgamma, float* dbeta, int N, int T, int H, int W,
                              sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (int e = stem_dims("sbl_stem_wgrad", N, T, H, W)) return e;
    SBL_REQUIRE(x && conv_out && dpooled && argmax && mean && invstd && gamma && beta && sums && dw && dgamma && dbeta,
                "sbl_stem_wgrad: null pointer");
    const int Ho = H / 2, Wo = W / 2, TY = sbl_cdiv(Ho, ST_TH), TX = sbl_cdiv(Wo, ST_TW);
    const long ntiles = (long)N * T * TY * TX;
    SBL_REQUIRE(ntiles < (1L << 31), "sbl_stem_wgrad: too many tiles");
    SBL_HIP(hipMemsetAsync(dw, 0, sizeof(float) * 64 * ST_K, s));
    const int grid = (int)(ntiles < 768 ? ntiles : 768);   // 3 workgroups per CU (52 KB LDS each)
    if (g_sbl_prec && g_sbl_stem_wgrad_tr) {
        const int grid2 = (int)(ntiles < 512 ? ntiles : 512);   // 2 workgroups per CU (77 KB LDS each)
#define SBL_SWT_(NT)                                                                                                           \
    do {                                                                                                                       \
        const size_t lds = 2 * BfTerms<NT>::NPL * SW_PLANE + BfTerms<NT>::NPL * SW_DPLANE + 6 * 64 * sizeof(float);                                   \
        static bool set_[64] = {false};                                                                                        \
        int dev = 0;                                                                                                           \
        SBL_HIP(hipGetDevice(&dev));                                                                                           \
        if (!set_[dev & 63]) {                                                                                                 \
            SBL_HIP(hipFuncSetAttribute((const void*)stem_wgrad_tr_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
            set_[dev & 63] = true;                                                                                             \
        }                                                                                                                      \
        hipLaunchKernelGGL(stem_wgrad_tr_kernel<NT>, dim3(grid2), dim3(SW_THREADS), lds, s, x, conv_out, dpooled, argmax, mean, invstd, gamma, \
                           beta, sums, dw, dgamma, dbeta, N, T, H, W, Ho, Wo, Ho / 2, Wo / 2, TY, TX, (int)ntiles, sbl_next_stamp_slot(SBL_KID_STEM), g_sbl_stem_ablate); \
    } while (0)
        if (g_sbl_prec == 6) SBL_SWT_(6);
        else if (g_sbl_prec == 3) SBL_SWT_(3);
        else SBL_SWT_(1);
#undef SBL_SWT_
        SBL_LAUNCH_CHECK("sbl_stem_wgrad(bf16, transposed reads)");
        return 0;
    }
    if (g_sbl_prec) {
#define SBL_SWG_(NT) hipLaunchKernelGGL(stem_wgrad_bf_kernel<NT>, dim3(grid), dim3(256), 0, s, x, conv_out, dpooled, argmax, mean, invstd, gamma, \
                       beta, sums, dw, dgamma, dbeta, N, T, H, W, Ho, Wo, Ho / 2, Wo / 2, TY, TX, (int)ntiles, sbl_next_stamp_slot(SBL_KID_STEM))
        if (g_sbl_prec == 6) SBL_SWG_(6);
        else if (g_sbl_prec == 3) SBL_SWG_(3);
        else SBL_SWG_(1);
#undef SBL_SWG_
        SBL_LAUNCH_CHECK("sbl_stem_wgrad(bf16)");
        return 0;
    }
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(256), 0, s, x, conv_out, dpooled, argmax, mean, invstd, gamma,
                       beta, sums, dw, dgamma, dbeta, N, T, H, W, Ho, Wo, Ho / 2, Wo / 2, TY, TX, (int)ntiles, sbl_next_stamp_slot(SBL_KID_STEM));
    SBL_LAUNCH_CHECK("sbl_stem_wgrad");
    return 0;
}
