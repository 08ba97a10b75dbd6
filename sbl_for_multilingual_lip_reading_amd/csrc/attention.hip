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
                                                            uint64_t offset) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Qs = smem, *Ks = smem + AT_SZ, *Vs = smem + 2 * AT_SZ, *Ss = smem + 3 * AT_SZ;
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
}

__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ dout, long lddo, const float* __restrict__ q,
                                                            long ldq, const float* __restrict__ k, long ldk,
                                                            const float* __restrict__ v, long ldv, const float* __restrict__ p,
                                                            float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
                                                            float* __restrict__ dv, long lddv, int B, int H, SegDesc segs,
                                                            int Lk_fixed, float scale, uint32_t thresh, float keep_scale,
                                                            const uint64_t* __restrict__ seed, uint64_t offset) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Qs = smem, *Ks = smem + AT_SZ, *Vs = smem + 2 * AT_SZ, *Gs = smem + 3 * AT_SZ, *Ps = smem + 4 * AT_SZ,
          *Ds = smem + 5 * AT_SZ;   // Gs = dO, Ps = (dropped) P then dS, Ds = dP
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
// blocks back to back, each (H*B, L, Lk).  In backward with shared keys and nseg > 1, dk/dv must be zeroed by the
// caller (contributions are atomically accumulated).
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
    const size_t lds = sizeof(float) * 4 * AT_SZ;
    static bool attr_set[64] = {false};
    if (int e = at_attr((const void*)attention_fwd_kernel, lds, attr_set)) return e;
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(nseg * B * H), dim3(256), lds, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, o,
                       ldo, p_out, mask_kind, mask, B, H, d, Lk_fixed, scale, drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u,
                       1.f / (1.f - drop_p), seed, offset);
    SBL_LAUNCH_CHECK("sbl_attention_fwd");
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
    const size_t lds = sizeof(float) * 6 * AT_SZ;
    static bool attr_set[64] = {false};
    if (int e = at_attr((const void*)attention_bwd_kernel, lds, attr_set)) return e;
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(nseg * B * H), dim3(256), lds, (hipStream_t)stream, dout, lddo, q, ldq, k, ldk,
                       v, ldv, p, dq, lddq, dk, lddk, dv, lddv, B, H, d, Lk_fixed, scale,
                       drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset);
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
