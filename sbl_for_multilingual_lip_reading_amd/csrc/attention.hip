// ScaledDotProductAttention forward / backward (SBL/transformer/attention.py:63-83) for the sequence lengths
// of this model (T = 29..64 frames, <= 16 target tokens): one 256-thread workgroup per (batch, head) keeps
// Q, K, V (and dO, P, dP in backward) resident in LDS; QK^T, PV and the five backward products run on
// v_mfma_f32_32x32x2_f32 straight from LDS (row stride 65 floats: every operand read, straight or transposed,
// hits 32 distinct banks); the softmax row reductions are wavefront shuffles.  Heads are addressed in place as
// 64-wide column blocks of the (B, L, H*64) projection outputs, so the reference's permute/contiguous copies
// (attention.py:45-47,53-54) never happen.  P is written out head-major (H*B, Lq, Lk) like the reference's attn.
#include "sbl_common.h"

#define AT_LD 65
#define AT_SZ (64 * AT_LD)

// acc(32x32 tile at rows i0, cols j0) = sum_{k<K} A(i0+i,k) * B(k,j0+j),  A(i,k) = As[i*a_si + k*a_sk],
// B(k,j) = Bs[k*b_sk + j*b_sj].  K even.  Lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31].
__device__ __forceinline__ f32x16 lds_mma(const float* As, int a_si, int a_sk, const float* Bs, int b_sk, int b_sj,
                                          int i0, int j0, int K, int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = As + (i0 + (lane & 31)) * a_si + (lane >> 5) * a_sk;
    const float* bp = Bs + (lane >> 5) * b_sk + (j0 + (lane & 31)) * b_sj;
    for (int k = 0; k < K; k += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * a_sk], bp[k * b_sk], acc, 0, 0, 0);
    }
    return acc;
}

// load a (L x 64) head slice (row stride ld) into dst[r*65 + c], zero rows L..Lpad-1
__device__ __forceinline__ void load_head(float* dst, const float* __restrict__ src, long ld, int L, int Lpad, int tid) {
    for (int i = tid; i < Lpad * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < L) v = *reinterpret_cast<const float4*>(src + (long)r * ld + c);
        float* d = dst + r * AT_LD + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__device__ __forceinline__ bool at_masked(int mask_kind, const uint8_t* mask, int b, int i, int j, int Lq, int Lk) {
    if (mask_kind == 1) return j > i;
    if (mask_kind == 2) return mask[((long)b * Lq + i) * Lk + j] != 0;
    return false;
}

__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k,
                                                            long ldk, const float* __restrict__ v, long ldv,
                                                            float* __restrict__ o, long ldo, float* __restrict__ p_out,
                                                            int mask_kind, const uint8_t* __restrict__ mask, int B, int H,
                                                            SegDesc segs, int Lk_fixed, float scale, uint32_t thresh,
                                                            float keep_scale, const uint64_t* __restrict__ seed,
                                                            uint64_t offset, int sz, unsigned long long* stamp) {
    sbl_stamp_begin(stamp);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Qs = smem, *Ks = smem + sz, *Vs = smem + 2 * sz, *Ss = smem + 3 * sz;      // sz = rows actually needed x AT_LD
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (segment, batch row, head).  Self-attention (Lk_fixed == 0): keys are the segment's own rows;
    // cross-attention: every segment attends to the same (B, Lk_fixed) key/value rows.
    const int sidx = blockIdx.x / (B * H);
    const int b = (blockIdx.x / H) % B, h = blockIdx.x % H;
    const int Lq = segs.L[sidx], Lk = Lk_fixed > 0 ? Lk_fixed : Lq;
    const long qrow = segs.row_off[sidx] + (long)b * Lq;
    const long krow = Lk_fixed > 0 ? (long)b * Lk : qrow;
    const int Lqp = (Lq + 31) & ~31, Lkp = (Lk + 31) & ~31;
    load_head(Qs, q + qrow * ldq + h * 64, ldq, Lq, Lqp, tid);
    load_head(Ks, k + krow * ldk + h * 64, ldk, Lk, Lkp, tid);
    load_head(Vs, v + krow * ldv + h * 64, ldv, Lk, Lkp, tid);
    __syncthreads();
    // S = Q K^T * scale
    const int TQ = Lqp >> 5, TK = Lkp >> 5;
    for (int t = wave; t < TQ * TK; t += 4) {
        const int i0 = (t / TK) * 32, j0 = (t % TK) * 32;
        f32x16 acc = lds_mma(Qs, AT_LD, 1, Ks, 1, AT_LD, i0, j0, 64, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Ss[(i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * AT_LD + j0 + (lane & 31)] = acc[r] * scale;
    }
    __syncthreads();
    // softmax over keys, one wavefront per query row
    const uint64_t sd = thresh ? *seed : 0;
    const long pbase = segs.p_off[sidx] + ((long)h * B + b) * Lq * Lk;
    float* pg = p_out + pbase;
    for (int i = wave; i < Lqp; i += 4) {
        float pv = 0.f, pd = 0.f;
        if (i < Lq) {
            const bool valid = lane < Lk && !at_masked(mask_kind, mask, b, i, lane, Lq, Lk);
            const float s = valid ? Ss[i * AT_LD + lane] : -INFINITY;
            const float m = wave_max(s);
            const float e = valid ? __expf(s - m) : 0.f;
            const float sum = wave_sum(e);
            pv = sum > 0.f ? e / sum : 0.f;
            if (lane < Lk) pg[(long)i * Lk + lane] = pv;
            pd = pv;
            if (thresh)
                pd = sbl_keep(sd, offset, (uint64_t)pbase + (uint64_t)i * Lk + lane, thresh) ? pv * keep_scale : 0.f;
        }
        if (lane < Lkp) Ss[i * AT_LD + lane] = pd;
    }
    __syncthreads();
    // O = P V   (contraction over keys; rows >= Lk of P's columns / V are zero)
    float* og = o + qrow * ldo + h * 64;
    for (int t = wave; t < TQ * 2; t += 4) {
        const int i0 = (t >> 1) * 32, j0 = (t & 1) * 32;
        f32x16 acc = lds_mma(Ss, AT_LD, 1, Vs, AT_LD, 1, i0, j0, Lkp, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < Lq) og[(long)i * ldo + j0 + (lane & 31)] = acc[r];
        }
    }
    sbl_stamp_end(stamp);
}

__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ dout, long lddo, const float* __restrict__ q,
                                                            long ldq, const float* __restrict__ k, long ldk,
                                                            const float* __restrict__ v, long ldv, const float* __restrict__ p,
                                                            float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
                                                            float* __restrict__ dv, long lddv, int B, int H, SegDesc segs,
                                                            int Lk_fixed, float scale, uint32_t thresh, float keep_scale,
                                                            const uint64_t* __restrict__ seed, uint64_t offset, int sz, unsigned long long* stamp) {
    sbl_stamp_begin(stamp);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Qs = smem, *Ks = smem + sz, *Vs = smem + 2 * sz, *Gs = smem + 3 * sz, *Ps = smem + 4 * sz,
          *Ds = smem + 5 * sz;   // Gs = dO, Ps = (dropped) P then dS, Ds = dP; sz = rows actually needed x AT_LD
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sidx = blockIdx.x / (B * H);
    const int b = (blockIdx.x / H) % B, h = blockIdx.x % H;
    const int Lq = segs.L[sidx], Lk = Lk_fixed > 0 ? Lk_fixed : Lq;
    const long qrow = segs.row_off[sidx] + (long)b * Lq;
    const long krow = Lk_fixed > 0 ? (long)b * Lk : qrow;
    const int Lqp = (Lq + 31) & ~31, Lkp = (Lk + 31) & ~31;
    load_head(Qs, q + qrow * ldq + h * 64, ldq, Lq, Lqp, tid);
    load_head(Ks, k + krow * ldk + h * 64, ldk, Lk, Lkp, tid);
    load_head(Vs, v + krow * ldv + h * 64, ldv, Lk, Lkp, tid);
    load_head(Gs, dout + qrow * lddo + h * 64, lddo, Lq, Lqp, tid);
    const uint64_t sd = thresh ? *seed : 0;
    const long pbase = segs.p_off[sidx] + ((long)h * B + b) * Lq * Lk;
    const float* pg = p + pbase;
    for (int i = tid; i < Lqp * Lkp; i += 256) {   // Ps = P after dropout (zero padded)
        const int r = i / Lkp, c = i - r * Lkp;
        float val = 0.f;
        if (r < Lq && c < Lk) {
            val = pg[(long)r * Lk + c];
            if (thresh) val = sbl_keep(sd, offset, (uint64_t)pbase + (uint64_t)r * Lk + c, thresh) ? val * keep_scale : 0.f;
        }
        Ps[r * AT_LD + c] = val;
    }
    __syncthreads();
    const int TQ = Lqp >> 5, TK = Lkp >> 5;
    // dV[j][d] = sum_i Pd[i][j] dO[i][d]
    // cross-attention: several segments (decoder steps of one run) share the key/value rows, so their dK/dV
    // contributions are accumulated with float atomics into a zero-initialised buffer
    float* dvg = dv + krow * lddv + h * 64;
    const bool kv_atomic = Lk_fixed > 0 && segs.nseg > 1;
    for (int t = wave; t < TK * 2; t += 4) {
        const int i0 = (t >> 1) * 32, j0 = (t & 1) * 32;
        f32x16 acc = lds_mma(Ps, 1, AT_LD, Gs, AT_LD, 1, i0, j0, Lqp, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (j < Lk) {
                if (kv_atomic) atomicAdd(dvg + (long)j * lddv + j0 + (lane & 31), acc[r]);
                else dvg[(long)j * lddv + j0 + (lane & 31)] = acc[r];
            }
        }
    }
    // dPd[i][j] = sum_d dO[i][d] V[j][d]
    for (int t = wave; t < TQ * TK; t += 4) {
        const int i0 = (t / TK) * 32, j0 = (t % TK) * 32;
        f32x16 acc = lds_mma(Gs, AT_LD, 1, Vs, 1, AT_LD, i0, j0, 64, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Ds[(i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * AT_LD + j0 + (lane & 31)] = acc[r];
    }
    __syncthreads();
    // dS = P * (dP - rowsum(dP * P)) * scale, with dP = dPd * keep/(1-p);  Ps <- dS
    for (int i = wave; i < Lqp; i += 4) {
        float ds = 0.f;
        if (i < Lq) {
            float pv = 0.f, dp = 0.f;
            if (lane < Lk) {
                pv = pg[(long)i * Lk + lane];
                dp = Ds[i * AT_LD + lane];
                if (thresh)
                    dp = sbl_keep(sd, offset, (uint64_t)pbase + (uint64_t)i * Lk + lane, thresh) ? dp * keep_scale : 0.f;
            }
            const float dot = wave_sum(pv * dp);
            ds = pv * (dp - dot) * scale;
        }
        if (lane < Lkp) Ps[i * AT_LD + lane] = ds;
    }
    __syncthreads();
    // dQ[i][d] = sum_j dS[i][j] K[j][d]
    float* dqg = dq + qrow * lddq + h * 64;
    for (int t = wave; t < TQ * 2; t += 4) {
        const int i0 = (t >> 1) * 32, j0 = (t & 1) * 32;
        f32x16 acc = lds_mma(Ps, AT_LD, 1, Ks, AT_LD, 1, i0, j0, Lkp, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < Lq) dqg[(long)i * lddq + j0 + (lane & 31)] = acc[r];
        }
    }
    // dK[j][d] = sum_i dS[i][j] Q[i][d]
    float* dkg = dk + krow * lddk + h * 64;
    for (int t = wave; t < TK * 2; t += 4) {
        const int i0 = (t >> 1) * 32, j0 = (t & 1) * 32;
        f32x16 acc = lds_mma(Ps, 1, AT_LD, Qs, AT_LD, 1, i0, j0, Lqp, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = i0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (j < Lk) {
                if (kv_atomic) atomicAdd(dkg + (long)j * lddk + j0 + (lane & 31), acc[r]);
                else dkg[(long)j * lddk + j0 + (lane & 31)] = acc[r];
            }
        }
    }
    sbl_stamp_end(stamp);
}

// ------------------------------------------------------------------ decoder-sized problems: one wavefront each
// Lq <= 16 target tokens, Lk <= 32 keys (self-attention over a prefix, or cross-attention to 29 frames): the
// (b, head, segment) problem is so small that a 256-thread workgroup spends its time in barriers and padding.
// Here one wavefront owns one problem, operands go global -> registers -> v_mfma_f32_16x16x4_f32 (exact fp32,
// same rate as 32x32x2) with no LDS and no barrier.  Lane l = (n = l & 15, g = l >> 4).  The MFMA contraction
// slot kk = g may stand for any index as long as A and B agree, and so may the output column n, which lets every
// operand be fetched with 16-byte loads:
//   row fragment of X (16 rows x 64): lane (n, g) holds X[n][16c + 4g + e], c, e = 0..3  (4 float4 loads)
//   S^T tile t (keys 16t..16t+15) = K_t Q^T  ->  lane (n, g) reg r = S[i = n][j = 16t + 4g + r]      ("T layout")
//   S   tile t                    = Q K_t^T  ->  lane (n, g) reg r = S[i = 4g + r][j = 16t + n]      ("N layout")
//   a T-layout matrix is an A operand for contractions over keys j (P V, dS K) with B = M[j][4n + e] (float4),
//   an N-layout matrix (rows = keys on n after the operand swap) for contractions over queries i (P^T dO, dS^T Q).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct RowFrag {
    float v[4][4];
};
__device__ __forceinline__ void frag_load(RowFrag& f, const float* __restrict__ base, long ld, int row, bool ok, int g) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) t = *reinterpret_cast<const float4*>(base + (long)row * ld + 16 * c + 4 * g);
        f.v[c][0] = t.x; f.v[c][1] = t.y; f.v[c][2] = t.z; f.v[c][3] = t.w;
    }
}
// D = A_frag(rows on n) * B_frag(rows on n)^T over the 64-wide head dim: D[4g + r][n] = sum_d A[4g + r][d] B[n][d]
__device__ __forceinline__ f32x4 frag_dot(const RowFrag& a, const RowFrag& b) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = MFMA16(a.v[c][e], b.v[c][e], acc);
    return acc;
}
__device__ __forceinline__ void ld4(float* dst, const float* __restrict__ src, bool ok) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) t = *reinterpret_cast<const float4*>(src);
    dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
}
__device__ __forceinline__ float grp_sum_hi(float v) {   // across g (lanes n, n+16, n+32, n+48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float grp_max_hi(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float grp_sum_lo(float v) {   // across n inside a 16-lane group
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct SmallProb {
    int Lq, Lk, b, h, qoff;      // qoff: index of the tile's first query inside its sequence (query-tile mode), else 0
    long qrow, krow, pbase;
};
__device__ __forceinline__ SmallProb small_prob(int prob, int B, int H, const SegDesc& segs, int Lk_fixed) {
    SmallProb P;
    const int sidx = prob / (B * H);
    P.b = (prob / H) % B;
    P.h = prob % H;
    int Lq = segs.L[0], ro = segs.row_off[0], po = segs.p_off[0];
#pragma unroll
    for (int t = 1; t < SBL_MAX_SEG; ++t)
        if (t == sidx) { Lq = segs.L[t]; ro = segs.row_off[t]; po = segs.p_off[t]; }
    P.Lq = Lq;
    P.Lk = Lk_fixed > 0 ? Lk_fixed : Lq;
    P.qrow = ro + (long)P.b * Lq;
    P.krow = Lk_fixed > 0 ? (long)P.b * P.Lk : P.qrow;
    P.pbase = po + ((long)P.h * B + P.b) * Lq * P.Lk;
    P.qoff = 0;
    if (segs.qtile) {      // query tiles of one self-attention over qtile (= Lk_fixed) rows: all tiles read the sequence's own keys
        P.qoff = ro;
        P.qrow = (long)P.b * segs.qtile + ro;
        P.pbase = ((long)P.h * B + P.b) * segs.qtile * P.Lk + po;
    }
    return P;
}

__device__ __forceinline__ void attention_small_fwd_body(const float* __restrict__ q, long ldq, const float* __restrict__ k,
                                                         long ldk, const float* __restrict__ v, long ldv,
                                                         float* __restrict__ o, long ldo, float* __restrict__ p_out,
                                                         int causal, int B, int H, const SegDesc& segs, int Lk_fixed,
                                                         float scale, uint32_t thresh, float keep_scale,
                                                         const uint64_t* __restrict__ seed, uint64_t offset, int nprob,
                                                         unsigned long long* stamp = nullptr) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int prob = blockIdx.x * 4 + (threadIdx.x >> 6);
    sbl_stamp_begin(stamp);      // (thread 0 = wavefront 0 of the workgroup, which always has a problem)
    if (prob >= nprob) return;   // whole wavefront; the kernel has no barrier
    const SmallProb P = small_prob(prob, B, H, segs, Lk_fixed);
    const int Lq = P.Lq, Lk = P.Lk;
    const int NT = Lk > 16 ? 2 : 1;
    const float* qb = q + P.qrow * ldq + P.h * 64;
    const float* kb = k + P.krow * ldk + P.h * 64;
    const float* vb = v + P.krow * ldv + P.h * 64;
    RowFrag qf;
    frag_load(qf, qb, ldq, n, n < Lq, g);
    // S in the T layout: lane (n, g) reg r of tile t = S[i = n][j = 16t + 4g + r]
    float s[2][4];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t < NT) {
            RowFrag kf;
            frag_load(kf, kb, ldk, 16 * t + n, 16 * t + n < Lk, g);
            const f32x4 acc = frag_dot(kf, qf);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * g + r;
                const bool valid = j < Lk && !(causal && j > n + P.qoff);
                s[t][r] = valid ? acc[r] * scale : -INFINITY;
                m = fmaxf(m, s[t][r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = -INFINITY;
        }
    }
    m = grp_max_hi(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[t][r] = s[t][r] == -INFINITY ? 0.f : __expf(s[t][r] - m);
            sum += s[t][r];
        }
    sum = grp_sum_hi(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    const uint64_t sd = thresh ? *seed : 0;
    float* pg = p_out + P.pbase;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            float pv = s[t][r] * inv;
            if (n < Lq && j < Lk) {
                pg[(long)n * Lk + j] = pv;
                if (thresh) pv = sbl_keep(sd, offset, (uint64_t)P.pbase + (uint64_t)n * Lk + j, thresh) ? pv * keep_scale : 0.f;
            } else {
                pv = 0.f;
            }
            s[t][r] = pv;
        }
    // O = Pd V: contraction over keys; lane (n, g) supplies A = Pd[i = n][j(g)], B = V[j(g)][4n + e]
    f32x4 oacc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oacc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * g + r;
                float vv[4];
                ld4(vv, vb + (long)j * ldv + 4 * n, j < Lk);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[e] = MFMA16(s[t][r], vv[e], oacc[e]);
            }
        }
    }
    float* ob = o + P.qrow * ldo + P.h * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        if (i < Lq) *reinterpret_cast<float4*>(ob + (long)i * ldo + 4 * n) = make_float4(oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]);
    }
    sbl_stamp_end(stamp);
}
__global__ __launch_bounds__(256) void attention_small_fwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k,
                                                                  long ldk, const float* __restrict__ v, long ldv,
                                                                  float* __restrict__ o, long ldo, float* __restrict__ p_out,
                                                                  int causal, int B, int H, SegDesc segs, int Lk_fixed,
                                                                  float scale, uint32_t thresh, float keep_scale,
                                                                  const uint64_t* __restrict__ seed, uint64_t offset, int nprob,
                                                                  unsigned long long* stamp) {
    attention_small_fwd_body(q, ldq, k, ldk, v, ldv, o, ldo, p_out, causal, B, H, segs, Lk_fixed, scale, thresh, keep_scale, seed, offset,
                             nprob, stamp);
}
// Two same-shape problems in one launch (the two decoder directions; blockIdx.y picks the operand set).
struct AttFwdSet {
    const float* q;
    const float* k;
    const float* v;
    float* o;
    float* p_out;
    uint64_t offset;
};
__global__ __launch_bounds__(256) void attention_small2_fwd_kernel(AttFwdSet a0, AttFwdSet a1, long ldq, long ldk, long ldv, long ldo,
                                                                   int causal, int B, int H, SegDesc segs, int Lk_fixed, float scale,
                                                                   uint32_t thresh, float keep_scale,
                                                                   const uint64_t* __restrict__ seed, int nprob) {
    const AttFwdSet& a = blockIdx.y ? a1 : a0;
    attention_small_fwd_body(a.q, ldq, a.k, ldk, a.v, ldv, a.o, ldo, a.p_out, causal, B, H, segs, Lk_fixed, scale, thresh, keep_scale, seed,
                             a.offset, nprob);
}

// SHARED_KV: cross-attention of a run with several segments.  All segments of one (batch, head) share the key /
// value rows, so they are the wavefronts of ONE workgroup (blockDim = 64 * nseg, grid = B * H): each adds its dK / dV
// tile into a 16 KB LDS image (ds_add_f32) and the workgroup stores the sums with plain 16-byte stores - no global
// atomics (they cost 33 / 70 us per launch at 2 / 5 segments against 12 us for one), no zero-filled output.
template <bool SHARED_KV>
__global__ __launch_bounds__(SHARED_KV ? 512 : 256) void attention_small_bwd_kernel(const float* __restrict__ dout, long lddo, const float* __restrict__ q,
                                                                  long ldq, const float* __restrict__ k, long ldk,
                                                                  const float* __restrict__ v, long ldv, const float* __restrict__ p,
                                                                  float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
                                                                  float* __restrict__ dv, long lddv, int B, int H, SegDesc segs,
                                                                  int Lk_fixed, float scale, uint32_t thresh, float keep_scale,
                                                                  const uint64_t* __restrict__ seed, uint64_t offset, int nprob,
                                                                  unsigned long long* stamp) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    sbl_stamp_begin(stamp);
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];      // SHARED_KV: [wavefront][dV | dK][32][64] partial tiles
    float* s_kv = s_dyn + (threadIdx.x >> 6) * 4096;                    // this wavefront's slice
    // SHARED_KV: wavefront w handles segments w, w + nw, ... of this (batch, head) one after the other and sums their dK / dV
    // tiles in its own LDS slice (up to 16 segments on 8 wavefronts: all 16 decoder steps in ONE launch, no second launch and
    // no add kernel for the halves)
    const int nwv = blockDim.x >> 6;
  for (int sidx = (int)(threadIdx.x >> 6), pass = 0; pass == 0 || (SHARED_KV && sidx < segs.nseg); sidx += nwv, ++pass) {
    const int prob = SHARED_KV ? sidx * (B * H) + (int)blockIdx.x : (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (!SHARED_KV && prob >= nprob) return;
    const SmallProb P = small_prob(prob, B, H, segs, Lk_fixed);
    const int Lq = P.Lq, Lk = P.Lk;
    const int NT = Lk > 16 ? 2 : 1;
    const float* qb = q + P.qrow * ldq + P.h * 64;
    const float* kb = k + P.krow * ldk + P.h * 64;
    const float* vb = v + P.krow * ldv + P.h * 64;
    const float* gb = dout + P.qrow * lddo + P.h * 64;
    const float* pg = p + P.pbase;
    const uint64_t sd = thresh ? *seed : 0;
    const bool kv_atomic = !SHARED_KV && Lk_fixed > 0 && segs.nseg > 1;
    RowFrag gf;   // dO rows on n
    frag_load(gf, gb, lddo, n, n < Lq, g);
    // ---- pass T: dP and P in the T layout (i = n, j = 16t + 4g + r)  ->  dS_T, then dQ = dS K
    float dsT[2][4];
    {
        float pT[2][4], dpT[2][4];
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < NT) {
                RowFrag vf;
                frag_load(vf, vb, ldv, 16 * t + n, 16 * t + n < Lk, g);
                const f32x4 acc = frag_dot(vf, gf);      // [j = 4g + r][i = n]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * g + r;
                    float pv = 0.f, dp = 0.f;
                    if (n < Lq && j < Lk) {
                        pv = pg[(long)n * Lk + j];
                        dp = acc[r];
                        if (thresh) dp = sbl_keep(sd, offset, (uint64_t)P.pbase + (uint64_t)n * Lk + j, thresh) ? dp * keep_scale : 0.f;
                    }
                    pT[t][r] = pv;
                    dpT[t][r] = dp;
                    dot += pv * dp;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) pT[t][r] = dpT[t][r] = 0.f;
            }
        }
        dot = grp_sum_hi(dot);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dsT[t][r] = pT[t][r] * (dpT[t][r] - dot) * scale;
    }
    {
        f32x4 acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * g + r;
                    float kk[4];
                    ld4(kk, kb + (long)j * ldk + 4 * n, j < Lk);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = MFMA16(dsT[t][r], kk[e], acc[e]);
                }
            }
        }
        float* dqb = dq + P.qrow * lddq + P.h * 64;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r;
            if (i < Lq) *reinterpret_cast<float4*>(dqb + (long)i * lddq + 4 * n) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
    }
    // ---- pass N: the same quantities in the N layout (i = 4g + r, j = 16t + n)  ->  dV = Pd^T dO, dK = dS^T Q
    float pdN[2][4], dsN[2][4];
    {
        float pN[2][4], dpN[2][4];
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < NT) {
                RowFrag vf;
                frag_load(vf, vb, ldv, 16 * t + n, 16 * t + n < Lk, g);
                const f32x4 acc = frag_dot(gf, vf);      // [i = 4g + r][j = n]
                const int j = 16 * t + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * g + r;
                    float pv = 0.f, dp = 0.f, pd = 0.f;
                    if (i < Lq && j < Lk) {
                        pv = pg[(long)i * Lk + j];
                        dp = acc[r];
                        pd = pv;
                        if (thresh) {
                            const bool keep = sbl_keep(sd, offset, (uint64_t)P.pbase + (uint64_t)i * Lk + j, thresh);
                            dp = keep ? dp * keep_scale : 0.f;
                            pd = keep ? pv * keep_scale : 0.f;
                        }
                    }
                    pN[t][r] = pv;
                    dpN[t][r] = dp;
                    pdN[t][r] = pd;
                    dot[r] += pv * dp;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) pN[t][r] = dpN[t][r] = pdN[t][r] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dot[r] = grp_sum_lo(dot[r]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dsN[t][r] = pN[t][r] * (dpN[t][r] - dot[r]) * scale;
    }
    // contraction over queries: slot kk = g stands for i = 4g + r; B rows = dO / Q rows i, columns 4n + e
    float gq[4][4], qq[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        ld4(gq[r], gb + (long)i * lddo + 4 * n, i < Lq);
        ld4(qq[r], qb + (long)i * ldq + 4 * n, i < Lq);
    }
    float* dvb = dv + P.krow * lddv + P.h * 64;
    float* dkb = dk + P.krow * lddk + P.h * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t < NT) {
            f32x4 av[4], ak[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = ak[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    av[e] = MFMA16(pdN[t][r], gq[r][e], av[e]);     // A[row j = n][kk -> i] = Pd[i][j]
                    ak[e] = MFMA16(dsN[t][r], qq[r][e], ak[e]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * g + r;
                if (j < Lk) {
                    float* pv_ = dvb + (long)j * lddv + 4 * n;
                    float* pk_ = dkb + (long)j * lddk + 4 * n;
                    if (SHARED_KV) {      // (LDS float atomics into one shared image were measured 20 us slower at 5 segments)
                        float4 a4 = make_float4(av[0][r], av[1][r], av[2][r], av[3][r]);
                        float4 k4 = make_float4(ak[0][r], ak[1][r], ak[2][r], ak[3][r]);
                        if (pass) {        // this lane wrote the same addresses in its previous pass
                            const float4 pa = *reinterpret_cast<const float4*>(&s_kv[j * 64 + 4 * n]);
                            const float4 pk = *reinterpret_cast<const float4*>(&s_kv[2048 + j * 64 + 4 * n]);
                            a4.x += pa.x; a4.y += pa.y; a4.z += pa.z; a4.w += pa.w;
                            k4.x += pk.x; k4.y += pk.y; k4.z += pk.z; k4.w += pk.w;
                        }
                        *reinterpret_cast<float4*>(&s_kv[j * 64 + 4 * n]) = a4;
                        *reinterpret_cast<float4*>(&s_kv[2048 + j * 64 + 4 * n]) = k4;
                    } else if (kv_atomic) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            atomicAdd(pv_ + e, av[e][r]);
                            atomicAdd(pk_ + e, ak[e][r]);
                        }
                    } else {
                        *reinterpret_cast<float4*>(pv_) = make_float4(av[0][r], av[1][r], av[2][r], av[3][r]);
                        *reinterpret_cast<float4*>(pk_) = make_float4(ak[0][r], ak[1][r], ak[2][r], ak[3][r]);
                    }
                }
            }
        }
    }
  }
    if (SHARED_KV) {
        const int Lk = Lk_fixed;
        const int bb = ((int)blockIdx.x / H) % B, hh = (int)blockIdx.x % H;
        float* dvb = dv + (long)bb * Lk * lddv + hh * 64;
        float* dkb = dk + (long)bb * Lk * lddk + hh * 64;
        __syncthreads();
        const int nw = blockDim.x >> 6;
        for (int i = threadIdx.x; i < 2 * Lk * 16; i += blockDim.x) {
            const int which = i / (Lk * 16), rem = i - which * (Lk * 16);
            const int j = rem >> 4, c = (rem & 15) * 4;
            float4 val = *reinterpret_cast<const float4*>(&s_dyn[which * 2048 + j * 64 + c]);
            for (int w = 1; w < nw; ++w) {      // segment order: deterministic
                const float4 u = *reinterpret_cast<const float4*>(&s_dyn[w * 4096 + which * 2048 + j * 64 + c]);
                val.x += u.x; val.y += u.y; val.z += u.z; val.w += u.w;
            }
            float* dst = which ? dkb + (long)j * lddk + c : dvb + (long)j * lddv + c;
            *reinterpret_cast<float4*>(dst) = val;
        }
    }
    sbl_stamp_end(stamp);
}

// Self-attention over 17..32 rows (the encoder's 29 frames): two query tiles of <= 16 rows per (batch, head), each a
// one-wavefront problem over all of the sequence's keys; in backward the tiles' dK / dV meet in LDS like the segments of a
// cross-attention run (SHARED_KV).  attention.py:72-83 at the encoder's size without the 256-thread / 50-100 KB workgroup.
static bool at_qtile_ok(const SegDesc& d, int Lk_fixed, int mask_kind) {
    return d.nseg == 1 && mask_kind != 2 && d.L[0] > 16 && d.L[0] <= 32 && Lk_fixed <= 32;
}
static SegDesc at_qtile_desc(int L, int Lk) {      // L query rows (two tiles), Lk keys
    SegDesc t;
    t.nseg = 2;
    t.qtile = L;
    for (int s = 0; s < SBL_MAX_SEG; ++s) {
        t.L[s] = s == 0 ? 16 : (s == 1 ? L - 16 : 0);
        t.row_off[s] = s < 2 ? 16 * s : 0;
        t.p_off[s] = s < 2 ? 16 * s * Lk : 0;
    }
    return t;
}
static bool at_small_ok(const SegDesc& d, int Lk_fixed, int mask_kind) {
    constexpr int enabled = 1;
    if (!enabled || mask_kind == 2 || Lk_fixed > 32) return false;
    for (int s = 0; s < d.nseg; ++s)
        if (d.L[s] > 16) return false;
    return true;
}

static int at_check(const char* who, int B, int H, const SegDesc& d, int Lk_fixed, long ldq, long ldk, long ldv, long ldo) {
    SBL_REQUIRE(B > 0 && H > 0 && Lk_fixed >= 0 && Lk_fixed <= 64, "%s: bad B=%d H=%d Lk=%d (Lk <= 64)", who, B, H, Lk_fixed);
    for (int s = 0; s < d.nseg; ++s)
        SBL_REQUIRE(d.L[s] >= 1 && d.L[s] <= 64, "%s: need 1 <= Lq,Lk <= 64 (segment %d has L=%d)", who, s, d.L[s]);
    SBL_REQUIRE((long)d.nseg * B * H < (1L << 30), "%s: too many workgroups", who);
    SBL_REQUIRE(ldq >= H * 64 && ldk >= H * 64 && ldv >= H * 64 && ldo >= H * 64, "%s: row stride smaller than H*64", who);
    SBL_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "%s: row strides must be multiples of 4 floats", who);
    return 0;
}

// LDS floats per staged matrix of the workgroup kernels: the longest padded side of any segment x the row stride.  (Sized by
// the problem, not by the 64-row maximum: at the encoder's 29 frames the backward kernel takes 50 KB instead of 100 KB and
// fits beside the other stream's GEMM workgroups - in the step its launches waited for a whole CU's LDS before.)
static int at_rows_sz(const SegDesc& d, int nseg, int Lk_fixed) {
    int m = Lk_fixed;
    for (int i = 0; i < nseg; ++i) m = d.L[i] > m ? d.L[i] : m;
    return ((m + 31) & ~31) * AT_LD;
}
static int at_attr(const void* fn, size_t lds, bool* flags) {
    int dev = 0;
    SBL_HIP(hipGetDevice(&dev));
    if (!flags[dev & 63]) {   // > 64 KB of dynamic LDS: raise the cap once per device
        SBL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        flags[dev & 63] = true;
    }
    return 0;
}

// Segmented form: seg_L = host array of nseg prefix lengths (1..16 entries).  Lk_fixed == 0: self-attention inside
// each segment (q/k/v rows of segment s start at sum_{t<s} B*seg_L[t]); Lk_fixed > 0: every segment's queries attend
// to the same (B, Lk_fixed) key/value rows (decoder cross-attention).  p_out holds the segments' probability
// blocks back to back, each (H*B, L, Lk).  In backward dq / dk / dv are always overwritten (with shared keys and
// several segments the contributions are summed in LDS by one workgroup per (batch, head), or, beyond 8 segments /
// 16 queries, accumulated with float atomics into a buffer the call zero-fills itself).
extern "C" int sbl_attention_seg_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o,
                                     long ldo, float* p_out, int mask_kind, const uint8_t* mask, int B, int H,
                                     const int* seg_L, int nseg, int Lk_fixed, float scale, float drop_p,
                                     const uint64_t* seed, uint64_t offset, sbl_stream_t stream) {
    SegDesc d;
    SBL_REQUIRE(sbl_make_segs(d, seg_L, nseg, B, H, Lk_fixed) > 0, "sbl_attention_seg_fwd: bad segment list (nseg=%d, 1..%d)", nseg, SBL_MAX_SEG);
    if (int e = at_check("sbl_attention_fwd", B, H, d, Lk_fixed, ldq, ldk, ldv, ldo)) return e;
    SBL_REQUIRE(q && k && v && o && p_out && sbl_aligned16(q) && sbl_aligned16(k) && sbl_aligned16(v), "sbl_attention_fwd: null/unaligned pointer");
    SBL_REQUIRE(mask_kind >= 0 && mask_kind <= 2 && (mask_kind != 2 || (mask && nseg == 1)), "sbl_attention_fwd: bad mask");
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_attention_fwd: bad dropout args");
    if (at_small_ok(d, Lk_fixed, mask_kind)) {
        SBL_REQUIRE(sbl_aligned16(o) && sbl_aligned16(q) && sbl_aligned16(k) && sbl_aligned16(v), "sbl_attention_fwd: unaligned pointer");
        const int nprob = nseg * B * H;
        hipLaunchKernelGGL(attention_small_fwd_kernel, dim3(sbl_cdiv(nprob, 4)), dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk, v,
                           ldv, o, ldo, p_out, mask_kind == 1, B, H, d, Lk_fixed, scale,
                           drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset, nprob, (unsigned long long*)nullptr);
        SBL_LAUNCH_CHECK("sbl_attention_fwd(small)");
        return 0;
    }
    if (at_qtile_ok(d, Lk_fixed, mask_kind) && sbl_aligned16(o)) {
        const int Lk = Lk_fixed > 0 ? Lk_fixed : d.L[0];
        const SegDesc t = at_qtile_desc(d.L[0], Lk);
        const int nprob = 2 * B * H;
        hipLaunchKernelGGL(attention_small_fwd_kernel, dim3(sbl_cdiv(nprob, 4)), dim3(256), 0, (hipStream_t)stream, q, ldq, k, ldk, v,
                           ldv, o, ldo, p_out, mask_kind == 1, B, H, t, Lk, scale,
                           drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset, nprob, sbl_next_stamp_slot(SBL_KID_ATTENTION));
        SBL_LAUNCH_CHECK("sbl_attention_fwd(query tiles)");
        return 0;
    }
    const int sz = at_rows_sz(d, nseg, Lk_fixed);
    const size_t lds = sizeof(float) * 4 * sz;
    static bool attr_set[64] = {false};
    if (int e = at_attr((const void*)attention_fwd_kernel, sizeof(float) * 4 * AT_SZ, attr_set)) return e;
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(nseg * B * H), dim3(256), lds, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, o,
                       ldo, p_out, mask_kind, mask, B, H, d, Lk_fixed, scale, drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u,
                       1.f / (1.f - drop_p), seed, offset, sz, sbl_next_stamp_slot(SBL_KID_ATTENTION));
    SBL_LAUNCH_CHECK("sbl_attention_fwd");
    return 0;
}

extern "C" int sbl_attention_seg2_fwd(const float* q0, const float* q1, long ldq, const float* k0, const float* k1, long ldk,
                                      const float* v0, const float* v1, long ldv, float* o0, float* o1, long ldo, float* p_out0,
                                      float* p_out1, int mask_kind, int B, int H, const int* seg_L, int nseg, int Lk_fixed,
                                      float scale, float drop_p, const uint64_t* seed, uint64_t offset0, uint64_t offset1,
                                      sbl_stream_t stream) {
    SegDesc d;
    SBL_REQUIRE(sbl_make_segs(d, seg_L, nseg, B, H, Lk_fixed) > 0, "sbl_attention_seg2_fwd: bad segment list (nseg=%d, 1..%d)", nseg, SBL_MAX_SEG);
    SBL_REQUIRE(mask_kind == 0 || mask_kind == 1, "sbl_attention_seg2_fwd: mask_kind %d (0 none, 1 causal)", mask_kind);
    if (!at_small_ok(d, Lk_fixed, mask_kind)) {     // sizes beyond the one-wavefront kernel: two plain launches
        if (int e = sbl_attention_seg_fwd(q0, ldq, k0, ldk, v0, ldv, o0, ldo, p_out0, mask_kind, nullptr, B, H, seg_L, nseg, Lk_fixed, scale, drop_p, seed, offset0, stream)) return e;
        return sbl_attention_seg_fwd(q1, ldq, k1, ldk, v1, ldv, o1, ldo, p_out1, mask_kind, nullptr, B, H, seg_L, nseg, Lk_fixed, scale, drop_p, seed, offset1, stream);
    }
    if (int e = at_check("sbl_attention_seg2_fwd", B, H, d, Lk_fixed, ldq, ldk, ldv, ldo)) return e;
    SBL_REQUIRE(q0 && q1 && k0 && k1 && v0 && v1 && o0 && o1 && p_out0 && p_out1, "sbl_attention_seg2_fwd: null pointer");
    SBL_REQUIRE(sbl_aligned16(q0) && sbl_aligned16(q1) && sbl_aligned16(k0) && sbl_aligned16(k1) && sbl_aligned16(v0) && sbl_aligned16(v1) &&
                    sbl_aligned16(o0) && sbl_aligned16(o1), "sbl_attention_seg2_fwd: unaligned pointer");
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_attention_seg2_fwd: bad dropout args");
    const int nprob = nseg * B * H;
    AttFwdSet a0{q0, k0, v0, o0, p_out0, offset0}, a1{q1, k1, v1, o1, p_out1, offset1};
    hipLaunchKernelGGL(attention_small2_fwd_kernel, dim3(sbl_cdiv(nprob, 4), 2), dim3(256), 0, (hipStream_t)stream, a0, a1, ldq, ldk, ldv,
                       ldo, mask_kind == 1, B, H, d, Lk_fixed, scale, drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p),
                       seed, nprob);
    SBL_LAUNCH_CHECK("sbl_attention_seg2_fwd");
    return 0;
}

extern "C" int sbl_attention_seg_bwd(const float* dout, long lddo, const float* q, long ldq, const float* k, long ldk,
                                     const float* v, long ldv, const float* p, float* dq, long lddq, float* dk, long lddk,
                                     float* dv, long lddv, int B, int H, const int* seg_L, int nseg, int Lk_fixed,
                                     float scale, float drop_p, const uint64_t* seed, uint64_t offset,
                                     sbl_stream_t stream) {
    SegDesc d;
    SBL_REQUIRE(sbl_make_segs(d, seg_L, nseg, B, H, Lk_fixed) > 0, "sbl_attention_seg_bwd: bad segment list (nseg=%d)", nseg);
    if (int e = at_check("sbl_attention_bwd", B, H, d, Lk_fixed, ldq, ldk, ldv, lddo)) return e;
    SBL_REQUIRE(lddq >= H * 64 && lddk >= H * 64 && lddv >= H * 64, "sbl_attention_bwd: gradient row stride smaller than H*64");
    SBL_REQUIRE(dout && q && k && v && p && dq && dk && dv && sbl_aligned16(dout) && sbl_aligned16(q) && sbl_aligned16(k) && sbl_aligned16(v),
                "sbl_attention_bwd: null/unaligned pointer");
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_attention_bwd: bad dropout args");
    if (at_small_ok(d, Lk_fixed, 0) && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && sbl_aligned16(dq) && sbl_aligned16(dk) &&
        sbl_aligned16(dv)) {
        const int nprob = nseg * B * H;
        constexpr int shared_kv = 1;
        if (Lk_fixed > 0 && nseg > 1 && !shared_kv) {      // atomics path accumulates: start from zero
            SBL_HIP(hipMemset2DAsync(dk, lddk * sizeof(float), 0, (size_t)H * 64 * sizeof(float), (size_t)B * Lk_fixed, (hipStream_t)stream));
            SBL_HIP(hipMemset2DAsync(dv, lddv * sizeof(float), 0, (size_t)H * 64 * sizeof(float), (size_t)B * Lk_fixed, (hipStream_t)stream));
        }
        const int nwv = nseg < 8 ? nseg : 8;      // 8 wavefronts = 2 per SIMD at this kernel's register use; beyond 8 segments each
                                                   // wavefront takes several (sbl_make_segs caps nseg at SBL_MAX_SEG = 16)
        if (shared_kv && Lk_fixed > 0 && nseg > 1) {
            static bool attr_set2[64] = {false};
            if (int e = at_attr((const void*)attention_small_bwd_kernel<true>, 8 * 16384, attr_set2)) return e;
        }
        if (shared_kv && Lk_fixed > 0 && nseg > 1)
            hipLaunchKernelGGL(attention_small_bwd_kernel<true>, dim3(B * H), dim3(64 * nwv), (size_t)nwv * 16384, (hipStream_t)stream, dout, lddo, q,
                               ldq, k, ldk, v, ldv, p, dq, lddq, dk, lddk, dv, lddv, B, H, d, Lk_fixed, scale,
                               drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset, nprob, (unsigned long long*)nullptr);
        else
            hipLaunchKernelGGL(attention_small_bwd_kernel<false>, dim3(sbl_cdiv(nprob, 4)), dim3(256), 0, (hipStream_t)stream, dout,
                               lddo, q, ldq, k, ldk, v, ldv, p, dq, lddq, dk, lddk, dv, lddv, B, H, d, Lk_fixed, scale,
                               drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset, nprob, (unsigned long long*)nullptr);
        SBL_LAUNCH_CHECK("sbl_attention_bwd(small)");
        return 0;
    }
    if (at_qtile_ok(d, Lk_fixed, 0) && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && sbl_aligned16(dq) && sbl_aligned16(dk) && sbl_aligned16(dv)) {
        const int Lk = Lk_fixed > 0 ? Lk_fixed : d.L[0];
        const SegDesc t = at_qtile_desc(d.L[0], Lk);
        static bool attr_set3[64] = {false};
        if (int e = at_attr((const void*)attention_small_bwd_kernel<true>, 8 * 16384, attr_set3)) return e;
        hipLaunchKernelGGL(attention_small_bwd_kernel<true>, dim3(B * H), dim3(128), (size_t)2 * 16384, (hipStream_t)stream, dout, lddo, q, ldq, k,
                           ldk, v, ldv, p, dq, lddq, dk, lddk, dv, lddv, B, H, t, Lk, scale,
                           drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset, 2 * B * H, sbl_next_stamp_slot(SBL_KID_ATTENTION));
        SBL_LAUNCH_CHECK("sbl_attention_bwd(query tiles)");
        return 0;
    }
    if (Lk_fixed > 0 && nseg > 1) {      // shared keys: the workgroup kernel accumulates dK / dV with float atomics
        SBL_HIP(hipMemset2DAsync(dk, lddk * sizeof(float), 0, (size_t)H * 64 * sizeof(float), (size_t)B * Lk_fixed, (hipStream_t)stream));
        SBL_HIP(hipMemset2DAsync(dv, lddv * sizeof(float), 0, (size_t)H * 64 * sizeof(float), (size_t)B * Lk_fixed, (hipStream_t)stream));
    }
    const int sz = at_rows_sz(d, nseg, Lk_fixed);
    const size_t lds = sizeof(float) * 6 * sz;
    static bool attr_set[64] = {false};
    if (int e = at_attr((const void*)attention_bwd_kernel, sizeof(float) * 6 * AT_SZ, attr_set)) return e;
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(nseg * B * H), dim3(256), lds, (hipStream_t)stream, dout, lddo, q, ldq, k, ldk,
                       v, ldv, p, dq, lddq, dk, lddk, dv, lddv, B, H, d, Lk_fixed, scale,
                       drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset, sz, sbl_next_stamp_slot(SBL_KID_ATTENTION));
    SBL_LAUNCH_CHECK("sbl_attention_bwd");
    return 0;
}

// Uniform form (one segment): Lq query rows and Lk key rows per batch entry, q/k/v possibly different tensors.
extern "C" int sbl_attention_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o,
                                 long ldo, float* p_out, int mask_kind, const uint8_t* mask, int B, int H, int Lq, int Lk,
                                 float scale, float drop_p, const uint64_t* seed, uint64_t offset, sbl_stream_t stream) {
    SBL_REQUIRE(Lq >= 1 && Lk >= 1 && Lq <= 64 && Lk <= 64, "sbl_attention_fwd: need 1 <= Lq,Lk <= 64 (got %d,%d)", Lq, Lk);
    return sbl_attention_seg_fwd(q, ldq, k, ldk, v, ldv, o, ldo, p_out, mask_kind, mask, B, H, &Lq, 1, Lk, scale, drop_p, seed,
                                 offset, stream);
}
extern "C" int sbl_attention_bwd(const float* dout, long lddo, const float* q, long ldq, const float* k, long ldk,
                                 const float* v, long ldv, const float* p, float* dq, long lddq, float* dk, long lddk,
                                 float* dv, long lddv, int B, int H, int Lq, int Lk, float scale, float drop_p,
                                 const uint64_t* seed, uint64_t offset, sbl_stream_t stream) {
    SBL_REQUIRE(Lq >= 1 && Lk >= 1 && Lq <= 64 && Lk <= 64, "sbl_attention_bwd: need 1 <= Lq,Lk <= 64 (got %d,%d)", Lq, Lk);
    return sbl_attention_seg_bwd(dout, lddo, q, ldq, k, ldk, v, ldv, p, dq, lddq, dk, lddk, dv, lddv, B, H, &Lq, 1, Lk, scale,
                                 drop_p, seed, offset, stream);
}
