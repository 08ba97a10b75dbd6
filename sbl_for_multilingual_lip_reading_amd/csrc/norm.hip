// HBM-bound normalisation kernels: trunk BatchNorm apply / backward (NHWC, float4 over channels),
// residual + LayerNorm forward / backward (one wavefront per 512-wide row, shuffle reductions),
// dropout, positional-encoding add.
#include "sbl_common.h"

// ------------------------------------------------------------------ BatchNorm apply (train or eval stats)
// y = [relu](gamma*(x-mean)*invstd + beta [+ res]);  video_frontend.py:30-39 (BasicBlock.forward)
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ y, long n4, int C4, int relu) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v.x - mu.x) * is.x * ga.x + be.x;
        o.y = (v.y - mu.y) * is.y * ga.y + be.y;
        o.z = (v.z - mu.z) * is.z * ga.z + be.z;
        o.w = (v.w - mu.w) * is.w * ga.w + be.w;
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// Training form with the BatchNorm "finalize" folded in: every thread keeps one channel quad for its whole grid-stride walk
// (the stride is a multiple of C/4, which divides 256), so it derives that quad's mean / invstd once from the (sum, sumsq)
// statistics of the producing convolution's epilogue - in double, exactly as bn_finalize_kernel does - and the first C/4
// threads of the grid also write what the separate launch wrote: save_mean / save_invstd for backward, the running
// statistics (momentum update, unbiased variance) and num_batches_tracked.  One launch per BatchNorm less (19 per step).
__global__ __launch_bounds__(256) void bn_apply_fwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                 const double* __restrict__ stats, long count,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float momentum, float eps, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ y,
                                                                 float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                 long long* __restrict__ nbt, long n4, int C4, int relu) {
    const long i0 = blockIdx.x * 256L + threadIdx.x;
    const int c = (int)(i0 % C4) * 4, C = C4 * 4;
    float mu[4], is[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double mean = stats[c + q] / (double)count;
        double var = stats[C + c + q] / (double)count - mean * mean;
        if (var < 0) var = 0;
        mu[q] = (float)mean;
        is[q] = (float)(1.0 / sqrt(var + (double)eps));
        if (i0 < C4) {
            save_mean[c + q] = mu[q];
            save_invstd[c + q] = is[q];
            if (running_mean) {
                const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
                running_mean[c + q] = (float)((1.0 - momentum) * running_mean[c + q] + momentum * mean);
                running_var[c + q] = (float)((1.0 - momentum) * running_var[c + q] + momentum * unbiased);
            }
        }
    }
    if (i0 == 0 && nbt) *nbt += 1;
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    const float4 be = *reinterpret_cast<const float4*>(beta + c);
    for (long i = i0; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        o.x = (v.x - mu[0]) * is[0] * ga.x + be.x;
        o.y = (v.y - mu[1]) * is[1] * ga.y + be.y;
        o.z = (v.z - mu[2]) * is[2] * ga.z + be.z;
        o.w = (v.w - mu[3]) * is[3] * ga.w + be.w;
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// backward pass 1: per-channel sum g and sum g*xhat, g = dy * (y > 0 if relu).
// A thread keeps one channel quad for its whole grid-stride walk (stride is a multiple of C4).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* __restrict__ sums,
                                                            long rows, int C, int relu, int rows_per_block,
                                                            float* __restrict__ part, int* __restrict__ ticket) {
    // blockDim = 256 = rg row-groups x C4 channel quads (C4 = C/4 divides 256)
    __shared__ float red[256][8];
    __shared__ int s_last;
    const int C4 = C / 4;
    const int rg = 256 / C4;                      // row groups per block
    const int q0 = threadIdx.x % C4, g0 = threadIdx.x / C4;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    {
        const int c = q0 * 4;
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        float sg[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0};
        auto fold = [&](float4 g, const float4& yy, const float4& v) {
            if (relu) {
                if (!(yy.x > 0.f)) g.x = 0.f;
                if (!(yy.y > 0.f)) g.y = 0.f;
                if (!(yy.z > 0.f)) g.z = 0.f;
                if (!(yy.w > 0.f)) g.w = 0.f;
            }
            sg[0] += g.x; sg[1] += g.y; sg[2] += g.z; sg[3] += g.w;
            sx[0] += g.x * (v.x - mu.x) * is.x;
            sx[1] += g.y * (v.y - mu.y) * is.y;
            sx[2] += g.z * (v.z - mu.z) * is.z;
            sx[3] += g.w * (v.w - mu.w) * is.w;
        };
        long r = r0 + g0;
        const long step = (long)rg * C;
        for (; r + 3L * rg < r1; r += 4L * rg) {      // 4 rows = 12 independent float4 loads in flight per lane
            const long o = r * C + c;
            float4 g[4], yy[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = *reinterpret_cast<const float4*>(dy + o + u * step);
                v[u] = *reinterpret_cast<const float4*>(x + o + u * step);
                yy[u] = relu ? *reinterpret_cast<const float4*>(y + o + u * step) : make_float4(1.f, 1.f, 1.f, 1.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(g[u], yy[u], v[u]);
        }
        for (; r < r1; r += rg) {
            const long o = r * C + c;
            const float4 g = *reinterpret_cast<const float4*>(dy + o);
            const float4 v = *reinterpret_cast<const float4*>(x + o);
            const float4 yy = relu ? *reinterpret_cast<const float4*>(y + o) : make_float4(1.f, 1.f, 1.f, 1.f);
            fold(g, yy, v);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[threadIdx.x][k] = sg[k];
            red[threadIdx.x][4 + k] = sx[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < C4) {
        float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < rg; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] += red[g * C4 + threadIdx.x][k];
        const int c = threadIdx.x * 4;
        if (part) {       // block partial -> workspace row [blockIdx.x][2C]
            float* pr = part + (long)blockIdx.x * 2 * C;
            *reinterpret_cast<float4*>(pr + c) = make_float4(t[0], t[1], t[2], t[3]);
            *reinterpret_cast<float4*>(pr + C + c) = make_float4(t[4], t[5], t[6], t[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(sums + c + k, (double)t[k]);
                atomicAdd(sums + C + c + k, (double)t[4 + k]);
            }
        }
    }
    if (!part) return;
    // the block that draws the last ticket sums the partials in block order (deterministic, no data atomics);
    // same release / acquire protocol as the split-K GEMM (mfma_gemm.h)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (tk == (int)gridDim.x - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // 2C/4 float4 columns x G block-groups = 256 lanes; group sums are combined through LDS in group order
    __shared__ double comb[256][4];
    const int Q = C / 2, G = 256 / Q;
    const int qq = threadIdx.x % Q, gi = threadIdx.x / Q;
    const int nb = (int)gridDim.x;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const float* pq = part + 4 * qq;
    int b = gi;
    for (; b + 3 * G < nb; b += 4 * G) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(pq + (long)(b + u * G) * 2 * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w; }
    }
    for (; b < nb; b += G) {
        const float4 v = *reinterpret_cast<const float4*>(pq + (long)b * 2 * C);
        a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) comb[threadIdx.x][k] = a[k];
    __syncthreads();
    if (gi == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t = 0.0;
            for (int g2 = 0; g2 < G; ++g2) t += comb[g2 * Q + qq][k];
            sums[4 * qq + k] = t;
        }
    }
}

// backward pass 2: dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dres = g; dgamma/dbeta from sums
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long n4, long rows, int C, int relu,
                                                           int accumulate) {
    const int C4 = C / 4;
    const double inv = 1.0 / (double)rows;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) {      // one writer: += is safe
            dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sums[c];
            dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sums[C + c];
        }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            const float4 yy = reinterpret_cast<const float4*>(y)[i];
            if (!(yy.x > 0.f)) g.x = 0.f;
            if (!(yy.y > 0.f)) g.y = 0.f;
            if (!(yy.z > 0.f)) g.z = 0.f;
            if (!(yy.w > 0.f)) g.w = 0.f;
        }
        if (dres) reinterpret_cast<float4*>(dres)[i] = g;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float mg0 = (float)(sums[c] * inv), mg1 = (float)(sums[c + 1] * inv), mg2 = (float)(sums[c + 2] * inv),
                    mg3 = (float)(sums[c + 3] * inv);
        const float mx0 = (float)(sums[C + c] * inv), mx1 = (float)(sums[C + c + 1] * inv),
                    mx2 = (float)(sums[C + c + 2] * inv), mx3 = (float)(sums[C + c + 3] * inv);
        float4 o;
        o.x = ga.x * is.x * (g.x - mg0 - (v.x - mu.x) * is.x * mx0);
        o.y = ga.y * is.y * (g.y - mg1 - (v.y - mu.y) * is.y * mx1);
        o.z = ga.z * is.z * (g.z - mg2 - (v.z - mu.z) * is.z * mx2);
        o.w = ga.w * is.w * (g.w - mg3 - (v.w - mu.w) * is.w * mx3);
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" int sbl_bn_apply_fwd(const float* x, const float* res, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float* y, long rows, int C, int relu,
                                sbl_stream_t stream) {
    SBL_REQUIRE(x && mean && invstd && gamma && beta && y && rows > 0 && C > 0 && C % 4 == 0, "sbl_bn_apply_fwd: bad args rows=%ld C=%d", rows, C);
    SBL_REQUIRE(sbl_aligned16(x) && sbl_aligned16(y) && (!res || sbl_aligned16(res)), "sbl_bn_apply_fwd: unaligned");
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, res, mean, invstd,
                       gamma, beta, y, n4, C / 4, relu);
    SBL_LAUNCH_CHECK("sbl_bn_apply_fwd");
    return 0;
}

extern "C" int sbl_bn_apply_fwd_stats(const float* x, const float* res, const double* stats, long count, float* running_mean,
                                      float* running_var, float momentum, float eps, const float* gamma, const float* beta,
                                      float* y, float* save_mean, float* save_invstd, int64_t* num_batches_tracked, long rows,
                                      int C, int relu, sbl_stream_t stream) {
    SBL_REQUIRE(x && stats && gamma && beta && y && save_mean && save_invstd && rows > 0 && count > 0 && C >= 4 && C % 4 == 0,
                "sbl_bn_apply_fwd_stats: bad args rows=%ld C=%d", rows, C);
    SBL_REQUIRE((C / 4) <= 256 && 256 % (C / 4) == 0, "sbl_bn_apply_fwd_stats: C=%d unsupported (C/4 must divide 256)", C);
    SBL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sbl_bn_apply_fwd_stats: running stats must both be set or both null");
    SBL_REQUIRE(sbl_aligned16(x) && sbl_aligned16(y) && (!res || sbl_aligned16(res)), "sbl_bn_apply_fwd_stats: unaligned");
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_apply_fwd_stats_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, res, stats, count,
                       running_mean, running_var, momentum, eps, gamma, beta, y, save_mean, save_invstd,
                       (long long*)num_batches_tracked, n4, C / 4, relu);
    SBL_LAUNCH_CHECK("sbl_bn_apply_fwd_stats");
    return 0;
}

extern "C" int sbl_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                 double* sums, long rows, int C, int relu, void* ws, long ws_bytes, sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(dy && x && mean && invstd && sums && rows > 0 && C >= 4 && C % 4 == 0 && (!relu || y), "sbl_bn_bwd_reduce: bad args");
    SBL_REQUIRE((C / 4) <= 256 && 256 % (C / 4) == 0, "sbl_bn_bwd_reduce: C=%d unsupported (C/4 must divide 256)", C);
    SBL_REQUIRE(!ws || (sbl_aligned16(ws) && ws_bytes >= 16384), "sbl_bn_bwd_reduce: workspace unaligned or < 16 KiB");
    // ws = the stream's GEMM workspace (int counters, all zero between launches, then fp32 slabs): block partials go
    // to the slabs and the last-arriving block reduces them; without a workspace every block ends with 2*C double
    // atomics on the same addresses, which serialise (measured 4x slower on the 22x22x64 layer)
    constexpr int max_blocks = 512;
    const int rg = 256 / (C / 4);
    long blocks = (rows + 4L * rg - 1) / (4L * rg);                 // >= 4 rows per lane
    if (blocks > max_blocks) blocks = max_blocks;
    if (ws && blocks > 131072 / (2 * C)) blocks = 131072 / (2 * C);     // the last block reads blocks*2C partials
    if (blocks < 1) blocks = 1;
    float* part = nullptr;
    int* ticket = nullptr;
    if (ws && ws_bytes >= 16384 + blocks * 2L * C * (long)sizeof(float)) {
        ticket = (int*)ws;
        part = (float*)((char*)ws + 16384);
    } else {
        SBL_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s));
    }
    const int rpb = (int)((rows + blocks - 1) / blocks);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((int)((rows + rpb - 1) / rpb)), dim3(256), 0, s, dy, y, x, mean, invstd,
                       sums, rows, C, relu, rpb, part, ticket);
    SBL_LAUNCH_CHECK("sbl_bn_bwd_reduce");
    return 0;
}

extern "C" int sbl_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                const float* gamma, const double* sums, float* dx, float* dres, float* dgamma,
                                float* dbeta, long rows, int C, int relu, int accumulate, sbl_stream_t stream) {
    SBL_REQUIRE(dy && x && mean && invstd && gamma && sums && dx && dgamma && dbeta && rows > 0 && C >= 4 && C % 4 == 0 && (!relu || y),
                "sbl_bn_bwd_apply: bad args");
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, dy, y, x, mean, invstd,
                       gamma, sums, dx, dres, dgamma, dbeta, n4, rows, C, relu, accumulate);
    SBL_LAUNCH_CHECK("sbl_bn_bwd_apply");
    return 0;
}

// ------------------------------------------------------------------ dropout
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                      uint32_t thresh, float scale, const uint64_t* __restrict__ seed,
                                                      uint64_t offset) {
    const uint64_t sd = *seed;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = sbl_keep(sd, offset, (uint64_t)i, thresh) ? x[i] * scale : 0.f;
}
__global__ void seed_bump_kernel(uint64_t* seed) { *seed = *seed * 6364136223846793005ull + 1442695040888963407ull; }

extern "C" int sbl_dropout(const float* x, float* y, long n, float p, const uint64_t* seed, uint64_t offset,
                           sbl_stream_t stream) {
    SBL_REQUIRE(x && y && seed && n > 0 && p >= 0.f && p < 1.f, "sbl_dropout: bad args n=%ld p=%f", n, p);
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, sbl_drop_thresh(p),
                       1.f / (1.f - p), seed, offset);
    SBL_LAUNCH_CHECK("sbl_dropout");
    return 0;
}
extern "C" int sbl_seed_bump(uint64_t* seed, sbl_stream_t stream) {
    SBL_REQUIRE(seed, "sbl_seed_bump: null");
    hipLaunchKernelGGL(seed_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed);
    SBL_LAUNCH_CHECK("sbl_seed_bump");
    return 0;
}

// ------------------------------------------------------------------ residual + LayerNorm (D = 512)
// One wavefront per row: 8 floats per lane (two float4), mean / variance by wave shuffles.
// attention.py:57-58, module.py:50-51, encoder.py:53-54 (torch LayerNorm: biased variance, eps inside sqrt)
__device__ __forceinline__ void add_layernorm_fwd_rows(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int M, float eps, uint32_t thresh,
                                                       float keep_scale, const uint64_t* __restrict__ seed,
                                                       uint64_t offset) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const long o = (long)row * 512;
    float v[8];
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(x + o + lane * 4);
    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(x + o + 256 + lane * 4);
    if (thresh) {   // fused dropout on x (same element indexing as dropout_kernel: mask is regenerated in backward)
        const uint64_t sd = *seed;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t idx = (uint64_t)o + (k < 4 ? 0 : 256) + lane * 4 + (k & 3);
            v[k] = sbl_keep(sd, offset, idx, thresh) ? v[k] * keep_scale : 0.f;
        }
    }
    if (res) {
        const float4 r0 = *reinterpret_cast<const float4*>(res + o + lane * 4);
        const float4 r1 = *reinterpret_cast<const float4*>(res + o + 256 + lane * 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    const float mu = wave_sum(s) * (1.f / 512.f);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) q += (v[k] - mu) * (v[k] - mu);
    const float rs = rsqrtf(wave_sum(q) * (1.f / 512.f) + eps);
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 4), g1 = *reinterpret_cast<const float4*>(gamma + 256 + lane * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + lane * 4), b1 = *reinterpret_cast<const float4*>(beta + 256 + lane * 4);
    float4 o0, o1;
    o0.x = (v[0] - mu) * rs * g0.x + b0.x; o0.y = (v[1] - mu) * rs * g0.y + b0.y;
    o0.z = (v[2] - mu) * rs * g0.z + b0.z; o0.w = (v[3] - mu) * rs * g0.w + b0.w;
    o1.x = (v[4] - mu) * rs * g1.x + b1.x; o1.y = (v[5] - mu) * rs * g1.y + b1.y;
    o1.z = (v[6] - mu) * rs * g1.z + b1.z; o1.w = (v[7] - mu) * rs * g1.w + b1.w;
    *reinterpret_cast<float4*>(y + o + lane * 4) = o0;
    *reinterpret_cast<float4*>(y + o + 256 + lane * 4) = o1;
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ y, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int M, float eps, uint32_t thresh,
                                                                float keep_scale, const uint64_t* __restrict__ seed,
                                                                uint64_t offset) {
    add_layernorm_fwd_rows(x, res, gamma, beta, y, mean, rstd, M, eps, thresh, keep_scale, seed, offset);
}
// Two same-shape problems in one launch (the two decoder directions; blockIdx.y picks the operand set).
struct LnFwdSet {
    const float* x;
    const float* res;
    const float* gamma;
    const float* beta;
    float* y;
    float* mean;
    float* rstd;
    uint64_t offset;
};
__global__ __launch_bounds__(256) void add_layernorm2_fwd_kernel(LnFwdSet a0, LnFwdSet a1, int M, float eps, uint32_t thresh,
                                                                 float keep_scale, const uint64_t* __restrict__ seed) {
    const LnFwdSet& a = blockIdx.y ? a1 : a0;
    add_layernorm_fwd_rows(a.x, a.res, a.gamma, a.beta, a.y, a.mean, a.rstd, M, eps, thresh, keep_scale, seed, a.offset);
}

// ------------------------------------------------------------------ last LayerNorm of a decoder layer + SBL fusion
// decoder.py:127-143: after the feed-forward sub-layer of layer n the two directions are fused, A' = A + flip(B), B' = 2B +
// flip(A) (time flip along each sequence's own prefix), and A', B' are the next layer's inputs.  A'[l] and B'[L-1-l] need
// exactly the same two LayerNorm rows - A[l] and B[L-1-l] - so one wavefront normalises those two rows (module.py:50-51,
// dropout on the sub-layer output included) and writes both fused rows: the LayerNorm outputs themselves are never
// stored (backward re-derives everything from the pre-norm sums, mean and rstd) and the separate fusion launch and its
// pass over two (rows, 512) tensors per layer disappear.
__device__ __forceinline__ void ln_row_regs(const LnFwdSet& a, int row, float eps, uint32_t thresh, float keep_scale, uint64_t sd,
                                            int lane, float (&out)[8]) {
    const long o = (long)row * 512;
    float v[8];
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(a.x + o + lane * 4);
    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(a.x + o + 256 + lane * 4);
    if (thresh) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t idx = (uint64_t)o + (k < 4 ? 0 : 256) + lane * 4 + (k & 3);
            v[k] = sbl_keep(sd, a.offset, idx, thresh) ? v[k] * keep_scale : 0.f;
        }
    }
    if (a.res) {
        const float4 r0 = *reinterpret_cast<const float4*>(a.res + o + lane * 4);
        const float4 r1 = *reinterpret_cast<const float4*>(a.res + o + 256 + lane * 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    const float mu = wave_sum(s) * (1.f / 512.f);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) q += (v[k] - mu) * (v[k] - mu);
    const float rs = rsqrtf(wave_sum(q) * (1.f / 512.f) + eps);
    const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + lane * 4), g1 = *reinterpret_cast<const float4*>(a.gamma + 256 + lane * 4);
    const float4 b0 = *reinterpret_cast<const float4*>(a.beta + lane * 4), b1 = *reinterpret_cast<const float4*>(a.beta + 256 + lane * 4);
    out[0] = (v[0] - mu) * rs * g0.x + b0.x; out[1] = (v[1] - mu) * rs * g0.y + b0.y;
    out[2] = (v[2] - mu) * rs * g0.z + b0.z; out[3] = (v[3] - mu) * rs * g0.w + b0.w;
    out[4] = (v[4] - mu) * rs * g1.x + b1.x; out[5] = (v[5] - mu) * rs * g1.y + b1.y;
    out[6] = (v[6] - mu) * rs * g1.z + b1.z; out[7] = (v[7] - mu) * rs * g1.w + b1.w;
    if (lane == 0) {
        a.mean[row] = mu;
        a.rstd[row] = rs;
    }
}
__global__ __launch_bounds__(256) void add_layernorm2_fusion_fwd_kernel(LnFwdSet a0, LnFwdSet a1, int B, SegDesc segs, int M, float eps,
                                                                        uint32_t thresh, float keep_scale,
                                                                        const uint64_t* __restrict__ seed) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;      // whole wavefront; no barrier in this kernel
    const int s = sbl_seg_of_row(segs, r, B);
    const int L = segs.L[s], l = (r - segs.row_off[s]) % L;
    const int rm = r + (L - 1 - 2 * l);      // the time-flipped position of the same sequence
    const uint64_t sd = thresh ? *seed : 0;
    float a[8], b[8];
    ln_row_regs(a0, r, eps, thresh, keep_scale, sd, lane, a);       // A[l]       (l2r direction)
    ln_row_regs(a1, rm, eps, thresh, keep_scale, sd, lane, b);      // B[L-1-l]   (r2l direction)
    float* xa = a0.y + (long)r * 512;       // A'[l]     = A[l] + B[L-1-l]
    float* xb = a1.y + (long)rm * 512;      // B'[L-1-l] = 2 B[L-1-l] + A[l]
    *reinterpret_cast<float4*>(xa + lane * 4) = make_float4(a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]);
    *reinterpret_cast<float4*>(xa + 256 + lane * 4) = make_float4(a[4] + b[4], a[5] + b[5], a[6] + b[6], a[7] + b[7]);
    *reinterpret_cast<float4*>(xb + lane * 4) = make_float4(2.f * b[0] + a[0], 2.f * b[1] + a[1], 2.f * b[2] + a[2], 2.f * b[3] + a[3]);
    *reinterpret_cast<float4*>(xb + 256 + lane * 4) = make_float4(2.f * b[4] + a[4], 2.f * b[5] + a[5], 2.f * b[6] + a[6], 2.f * b[7] + a[7]);
}

// dz = rstd*(gamma*dy - mean(gamma*dy) - xhat*mean(gamma*dy*xhat)); dgamma += dy*xhat, dbeta += dy (column sums:
// each wave walks its rows keeping 8 per-lane partials, LDS-combined per block, then float atomics)
#define SBL_LN_BWD_WAVES 8       // 512-thread workgroups: 8 waves x RW rows in flight per workgroup
__global__ __launch_bounds__(64 * SBL_LN_BWD_WAVES) void add_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ res, const float* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float* __restrict__ dz, float* __restrict__ dx_drop,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                                int rows_per_block, uint32_t thresh, float keep_scale,
                                                                const uint64_t* __restrict__ seed, uint64_t offset) {
    __shared__ float red[SBL_LN_BWD_WAVES][2][512];
    const uint64_t sd = thresh ? *seed : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 4), g1 = *reinterpret_cast<const float4*>(gamma + 256 + lane * 4);
    const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // each wave owns RW consecutive rows per trip and issues all of their loads before the first reduction, so
    // the (dependent) shuffle reductions of one row overlap the memory latency of the next
    constexpr int RW = 2;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int rb = r0 + wave * RW; rb < r1; rb += SBL_LN_BWD_WAVES * RW) {
        float v[RW][8], d[RW][8], mu[RW], rs[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int row = min(rb + j, r1 - 1);
            const long o = (long)row * 512;
            *reinterpret_cast<float4*>(v[j]) = *reinterpret_cast<const float4*>(x + o + lane * 4);
            *reinterpret_cast<float4*>(v[j] + 4) = *reinterpret_cast<const float4*>(x + o + 256 + lane * 4);
            *reinterpret_cast<float4*>(d[j]) = *reinterpret_cast<const float4*>(dy + o + lane * 4);
            *reinterpret_cast<float4*>(d[j] + 4) = *reinterpret_cast<const float4*>(dy + o + 256 + lane * 4);
            mu[j] = mean[row];
            rs[j] = rstd[row];
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int row = rb + j;
            if (row >= r1) break;
            const long o = (long)row * 512;
            bool keep[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                keep[k] = true;
                if (thresh) {
                    keep[k] = sbl_keep(sd, offset, (uint64_t)o + (k < 4 ? 0 : 256) + lane * 4 + (k & 3), thresh);
                    v[j][k] = keep[k] ? v[j][k] * keep_scale : 0.f;
                }
            }
            if (res) {
                const float4 q0 = *reinterpret_cast<const float4*>(res + o + lane * 4);
                const float4 q1 = *reinterpret_cast<const float4*>(res + o + 256 + lane * 4);
                v[j][0] += q0.x; v[j][1] += q0.y; v[j][2] += q0.z; v[j][3] += q0.w;
                v[j][4] += q1.x; v[j][5] += q1.y; v[j][6] += q1.z; v[j][7] += q1.w;
            }
            float s1 = 0.f, s2 = 0.f, xh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xh[k] = (v[j][k] - mu[j]) * rs[j];
                const float gd = ga[k] * d[j][k];
                s1 += gd;
                s2 += gd * xh[k];
                dg[k] += d[j][k] * xh[k];
                db[k] += d[j][k];
            }
            s1 = wave_sum(s1) * (1.f / 512.f);
            s2 = wave_sum(s2) * (1.f / 512.f);
            float out[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] = rs[j] * (ga[k] * d[j][k] - s1 - xh[k] * s2);
            *reinterpret_cast<float4*>(dz + o + lane * 4) = *reinterpret_cast<float4*>(out);
            *reinterpret_cast<float4*>(dz + o + 256 + lane * 4) = *reinterpret_cast<float4*>(out + 4);
            if (dx_drop) {   // gradient w.r.t. the pre-dropout x
#pragma unroll
                for (int k = 0; k < 8; ++k) out[k] = keep[k] ? out[k] * keep_scale : 0.f;
                *reinterpret_cast<float4*>(dx_drop + o + lane * 4) = *reinterpret_cast<float4*>(out);
                *reinterpret_cast<float4*>(dx_drop + o + 256 + lane * 4) = *reinterpret_cast<float4*>(out + 4);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = (k < 4 ? 0 : 256) + lane * 4 + (k & 3);
        red[wave][0][c] = dg[k];
        red[wave][1][c] = db[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64 * SBL_LN_BWD_WAVES) {      // the 2 x 512 column sums: one float atomic each
        const int which = i >> 9, c = i & 511;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SBL_LN_BWD_WAVES; ++w) t += red[w][which][c];
        atomicAdd((which ? dbeta : dgamma) + c, t);
    }
}

extern "C" int sbl_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                     float* mean, float* rstd, int M, int D, float eps, float drop_p,
                                     const uint64_t* seed, uint64_t offset, sbl_stream_t stream) {
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_add_layernorm_fwd: bad dropout args");
    SBL_REQUIRE(D == 512, "sbl_add_layernorm_fwd: D=%d (only d_model=512 is built: optimizer.py:8, decoder.py:59)", D);
    SBL_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0, "sbl_add_layernorm_fwd: bad args");
    SBL_REQUIRE(sbl_aligned16(x) && sbl_aligned16(y) && (!res || sbl_aligned16(res)) && sbl_aligned16(gamma) && sbl_aligned16(beta), "sbl_add_layernorm_fwd: unaligned");
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3(sbl_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, res, gamma,
                       beta, y, mean, rstd, M, eps, drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed,
                       offset);
    SBL_LAUNCH_CHECK("sbl_add_layernorm_fwd");
    return 0;
}

extern "C" int sbl_add_layernorm2_fwd(const float* x0, const float* x1, const float* res0, const float* res1, const float* gamma0,
                                      const float* gamma1, const float* beta0, const float* beta1, float* y0, float* y1,
                                      float* mean0, float* mean1, float* rstd0, float* rstd1, int M, int D, float eps, float drop_p,
                                      const uint64_t* seed, uint64_t offset0, uint64_t offset1, sbl_stream_t stream) {
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_add_layernorm2_fwd: bad dropout args");
    SBL_REQUIRE(D == 512, "sbl_add_layernorm2_fwd: D=%d (this build is specialised for d_model=512)", D);
    SBL_REQUIRE(x0 && x1 && gamma0 && gamma1 && beta0 && beta1 && y0 && y1 && mean0 && mean1 && rstd0 && rstd1 && M > 0 && (!res0 == !res1),
                "sbl_add_layernorm2_fwd: bad args");
    SBL_REQUIRE(sbl_aligned16(x0) && sbl_aligned16(x1) && sbl_aligned16(y0) && sbl_aligned16(y1) && (!res0 || (sbl_aligned16(res0) && sbl_aligned16(res1))) &&
                    sbl_aligned16(gamma0) && sbl_aligned16(gamma1) && sbl_aligned16(beta0) && sbl_aligned16(beta1), "sbl_add_layernorm2_fwd: unaligned");
    LnFwdSet a0{x0, res0, gamma0, beta0, y0, mean0, rstd0, offset0}, a1{x1, res1, gamma1, beta1, y1, mean1, rstd1, offset1};
    hipLaunchKernelGGL(add_layernorm2_fwd_kernel, dim3(sbl_cdiv(M, 4), 2), dim3(256), 0, (hipStream_t)stream, a0, a1, M, eps,
                       drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed);
    SBL_LAUNCH_CHECK("sbl_add_layernorm2_fwd");
    return 0;
}

/* LayerNorm(dropout(x_d) + res_d) of both decoder directions followed by the SBL cross-direction fusion, in one launch:
 * xn0 = A + flip(B), xn1 = 2B + flip(A) with A / B the two LayerNorm outputs (never stored); mean / rstd as in
 * sbl_add_layernorm2_fwd.  Rows are a ragged stage (B sequences per segment of length seg_L[s]). */
extern "C" int sbl_add_layernorm2_fusion_fwd(const float* x0, const float* x1, const float* res0, const float* res1, const float* gamma0,
                                             const float* gamma1, const float* beta0, const float* beta1, float* xn0, float* xn1,
                                             float* mean0, float* mean1, float* rstd0, float* rstd1, int B, const int* seg_L, int nseg,
                                             int D, float eps, float drop_p, const uint64_t* seed, uint64_t offset0, uint64_t offset1,
                                             sbl_stream_t stream) {
    SegDesc d;
    const long rows = sbl_make_segs(d, seg_L, nseg, B, 1, 1);
    SBL_REQUIRE(rows > 0 && rows < (1L << 30), "sbl_add_layernorm2_fusion_fwd: bad segment list");
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_add_layernorm2_fusion_fwd: bad dropout args");
    SBL_REQUIRE(D == 512, "sbl_add_layernorm2_fusion_fwd: D=%d (this build is specialised for d_model=512)", D);
    SBL_REQUIRE(x0 && x1 && gamma0 && gamma1 && beta0 && beta1 && xn0 && xn1 && mean0 && mean1 && rstd0 && rstd1 && (!res0 == !res1),
                "sbl_add_layernorm2_fusion_fwd: bad args");
    SBL_REQUIRE(xn0 != x0 && xn0 != x1 && xn1 != x0 && xn1 != x1 && xn0 != res0 && xn0 != res1 && xn1 != res0 && xn1 != res1 && xn0 != xn1,
                "sbl_add_layernorm2_fusion_fwd: outputs must not alias inputs (time flip)");
    SBL_REQUIRE(sbl_aligned16(x0) && sbl_aligned16(x1) && sbl_aligned16(xn0) && sbl_aligned16(xn1) && (!res0 || (sbl_aligned16(res0) && sbl_aligned16(res1))) &&
                    sbl_aligned16(gamma0) && sbl_aligned16(gamma1) && sbl_aligned16(beta0) && sbl_aligned16(beta1), "sbl_add_layernorm2_fusion_fwd: unaligned");
    LnFwdSet a0{x0, res0, gamma0, beta0, xn0, mean0, rstd0, offset0}, a1{x1, res1, gamma1, beta1, xn1, mean1, rstd1, offset1};
    hipLaunchKernelGGL(add_layernorm2_fusion_fwd_kernel, dim3(sbl_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a0, a1, B, d, (int)rows, eps,
                       drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed);
    SBL_LAUNCH_CHECK("sbl_add_layernorm2_fusion_fwd");
    return 0;
}

extern "C" int sbl_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma,
                                     const float* mean, const float* rstd, float* dz, float* dx_drop, float* dgamma,
                                     float* dbeta, int M, int D, float drop_p, const uint64_t* seed, uint64_t offset,
                                     sbl_stream_t stream) {
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_add_layernorm_bwd: bad dropout args");
    SBL_REQUIRE(!dx_drop || sbl_aligned16(dx_drop), "sbl_add_layernorm_bwd: dx_drop unaligned");
    SBL_REQUIRE(D == 512, "sbl_add_layernorm_bwd: D=%d", D);
    SBL_REQUIRE(dy && x && gamma && mean && rstd && dz && dgamma && dbeta && M > 0, "sbl_add_layernorm_bwd: bad args");
    SBL_REQUIRE(sbl_aligned16(dy) && sbl_aligned16(x) && sbl_aligned16(dz) && (!res || sbl_aligned16(res)) && sbl_aligned16(gamma), "sbl_add_layernorm_bwd: unaligned");
    // few, fat workgroups (each ends with 1024 float atomics on the same dgamma / dbeta words) of 8 waves: 16 rows in
    // flight per workgroup, two trips each at the stage-batched decoder's 4352 rows (136 workgroups).  256-thread
    // workgroups capped at 128 kept 364 waves in flight on the whole chip (28.6 us per launch in the step); 1024-thread
    // ones do not fit beside the other stream's GEMM workgroups (6 x 4 waves per CU) and wait for a CU to drain (58.8 us).
    constexpr int rows_trip = SBL_LN_BWD_WAVES * 2;
    int blocks = sbl_cdiv(M, 2 * rows_trip);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int rpb = sbl_cdiv(sbl_cdiv(M, blocks), rows_trip) * rows_trip;
    hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3(sbl_cdiv(M, rpb)), dim3(64 * SBL_LN_BWD_WAVES), 0, (hipStream_t)stream, dy, x, res,
                       gamma, mean, rstd, dz, dx_drop, dgamma, dbeta, M, rpb, drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u,
                       1.f / (1.f - drop_p), seed, offset);
    SBL_LAUNCH_CHECK("sbl_add_layernorm_bwd");
    return 0;
}

// ------------------------------------------------------------------ y = x + pe[row % L]
__global__ __launch_bounds__(256) void add_pe_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                     float* __restrict__ y, long n4, int L, int D4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const int l = (int)((i / D4) % L);
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        const float4 p = reinterpret_cast<const float4*>(pe)[(long)l * D4 + c];
        reinterpret_cast<float4*>(y)[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}
extern "C" int sbl_add_pe(const float* x, const float* pe, float* y, int B, int L, int D, sbl_stream_t stream) {
    SBL_REQUIRE(x && pe && y && B > 0 && L > 0 && D > 0 && D % 4 == 0, "sbl_add_pe: bad args");
    SBL_REQUIRE(sbl_aligned16(x) && sbl_aligned16(pe) && sbl_aligned16(y), "sbl_add_pe: unaligned");
    const long n4 = (long)B * L * (D / 4);
    hipLaunchKernelGGL(add_pe_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, pe, y, n4, L, D / 4);
    SBL_LAUNCH_CHECK("sbl_add_pe");
    return 0;
}
