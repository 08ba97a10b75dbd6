// Small decoder-side kernels: embedding + positional encoding, SBL cross-direction fusion, greedy /
// teacher-forced token feedback (kept on the device: no host sync inside the 16-step loop), label-smoothed
// cross entropy, fused Adam.
#include "sbl_common.h"

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// ------------------------------------------------------------------ embedding + PE: decoder.py:116-120
// Ragged form: segment s = decoder step with prefix length L[s]; its rows are (b, l), l < L[s], and every segment
// reads the same token buffer tok[b*ldt + l] (a step's prefix is the first L tokens of the final sequence).
__global__ __launch_bounds__(256) void embed_pe_fwd_kernel(const int64_t* __restrict__ tok, long ldt,
                                                           const float* __restrict__ emb, const float* __restrict__ pe,
                                                           float* __restrict__ out, int B, SegDesc segs, long rows, int D4,
                                                           int V) {
    const long n4 = rows * D4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const int r = (int)(i / D4);
        const int s = sbl_seg_of_row(segs, r, B);
        const int L = segs.L[s], rr = r - segs.row_off[s];
        const int b = rr / L, l = rr - b * L;
        long t = tok[(long)b * ldt + l];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);   // ids are produced on-device from argmax/gold: always in range
        const float4 e = reinterpret_cast<const float4*>(emb)[t * D4 + c];
        const float4 p = reinterpret_cast<const float4*>(pe)[(long)l * D4 + c];
        reinterpret_cast<float4*>(out)[i] = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tok, long ldt, const float* __restrict__ dy,
                                                        float* __restrict__ demb, int B, SegDesc segs, long rows, int D, int V) {
    const long n = rows * D;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D);
        const int r = (int)(i / D);
        const int s = sbl_seg_of_row(segs, r, B);
        const int L = segs.L[s], rr = r - segs.row_off[s];
        const int b = rr / L, l = rr - b * L;
        long t = tok[(long)b * ldt + l];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);
        atomicAdd(demb + t * D + c, dy[i]);
    }
}
extern "C" int sbl_embed_pe_seg_fwd(const int64_t* tok, long ldt, const float* emb, const float* pe, float* out, int B,
                                    const int* seg_L, int nseg, int D, int V, sbl_stream_t stream) {
    SegDesc d;
    const long rows = sbl_make_segs(d, seg_L, nseg, B, 1, 1);
    SBL_REQUIRE(rows > 0, "sbl_embed_pe_seg_fwd: bad segment list");
    SBL_REQUIRE(tok && emb && pe && out && B > 0 && D > 0 && D % 4 == 0 && V > 0, "sbl_embed_pe_fwd: bad args");
    for (int s = 0; s < nseg; ++s) SBL_REQUIRE(ldt >= seg_L[s], "sbl_embed_pe_fwd: token row shorter than prefix %d", seg_L[s]);
    SBL_REQUIRE(sbl_aligned16(emb) && sbl_aligned16(pe) && sbl_aligned16(out), "sbl_embed_pe_fwd: unaligned");
    hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3(ew_grid(rows * D / 4)), dim3(256), 0, (hipStream_t)stream, tok, ldt, emb, pe,
                       out, B, d, rows, D / 4, V);
    SBL_LAUNCH_CHECK("sbl_embed_pe_fwd");
    return 0;
}
extern "C" int sbl_embed_seg_bwd(const int64_t* tok, long ldt, const float* dy, float* demb, int B, const int* seg_L, int nseg,
                                 int D, int V, sbl_stream_t stream) {
    SegDesc d;
    const long rows = sbl_make_segs(d, seg_L, nseg, B, 1, 1);
    SBL_REQUIRE(rows > 0, "sbl_embed_seg_bwd: bad segment list");
    SBL_REQUIRE(tok && dy && demb && B > 0 && D > 0 && V > 0, "sbl_embed_bwd: bad args");
    for (int s = 0; s < nseg; ++s) SBL_REQUIRE(ldt >= seg_L[s], "sbl_embed_bwd: token row shorter than prefix %d", seg_L[s]);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(ew_grid(rows * D)), dim3(256), 0, (hipStream_t)stream, tok, ldt, dy, demb, B, d,
                       rows, D, V);
    SBL_LAUNCH_CHECK("sbl_embed_bwd");
    return 0;
}
extern "C" int sbl_embed_pe_fwd(const int64_t* tok, long ldt, const float* emb, const float* pe, float* out, int B, int L,
                                int D, int V, sbl_stream_t stream) {
    return sbl_embed_pe_seg_fwd(tok, ldt, emb, pe, out, B, &L, 1, D, V, stream);
}
extern "C" int sbl_embed_bwd(const int64_t* tok, long ldt, const float* dy, float* demb, int B, int L, int D, int V,
                             sbl_stream_t stream) {
    return sbl_embed_seg_bwd(tok, ldt, dy, demb, B, &L, 1, D, V, stream);
}

// ------------------------------------------------------------------ SBL fusion: decoder.py:132-143,160-164
// fwd: A' = A + flip(B), B' = 2B + flip(A).  bwd (adjoint): dA = dA' + flip(dB'), dB = flip(dA') + 2 dB'.
// The flip runs along each sequence's own prefix axis (per segment, per batch row).
template <bool BWD>
__global__ __launch_bounds__(256) void fusion_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ a2, float* __restrict__ b2, int B, SegDesc segs,
                                                     long rows, int D4) {
    const long n4 = rows * D4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const int r = (int)(i / D4);
        const int s = sbl_seg_of_row(segs, r, B);
        const int L = segs.L[s], rr = r - segs.row_off[s];
        const int l = rr % L;
        const long f = ((long)r + (L - 1 - 2 * l)) * D4 + c;   // time-flipped position inside the same sequence
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        const float4 xf = reinterpret_cast<const float4*>(a)[f], yf = reinterpret_cast<const float4*>(b)[f];
        float4 o1, o2;
        if (!BWD) {
            o1 = make_float4(x.x + yf.x, x.y + yf.y, x.z + yf.z, x.w + yf.w);
            o2 = make_float4(2.f * y.x + xf.x, 2.f * y.y + xf.y, 2.f * y.z + xf.z, 2.f * y.w + xf.w);
        } else {   // a = dA', b = dB'
            o1 = make_float4(x.x + yf.x, x.y + yf.y, x.z + yf.z, x.w + yf.w);
            o2 = make_float4(xf.x + 2.f * y.x, xf.y + 2.f * y.y, xf.z + 2.f * y.z, xf.w + 2.f * y.w);
        }
        reinterpret_cast<float4*>(a2)[i] = o1;
        reinterpret_cast<float4*>(b2)[i] = o2;
    }
}
static int fusion_common(const char* who, const float* a, const float* b, float* a2, float* b2, int B, const int* seg_L,
                         int nseg, int D, bool bwd, sbl_stream_t stream) {
    SegDesc d;
    const long rows = sbl_make_segs(d, seg_L, nseg, B, 1, 1);
    SBL_REQUIRE(rows > 0, "%s: bad segment list", who);
    SBL_REQUIRE(a && b && a2 && b2 && B > 0 && D > 0 && D % 4 == 0, "%s: bad args", who);
    SBL_REQUIRE(a2 != a && a2 != b && b2 != a && b2 != b && a2 != b2, "%s: outputs must not alias inputs (time flip)", who);
    SBL_REQUIRE(sbl_aligned16(a) && sbl_aligned16(b) && sbl_aligned16(a2) && sbl_aligned16(b2), "%s: unaligned", who);
    const long n4 = rows * D / 4;
    if (bwd) hipLaunchKernelGGL(fusion_kernel<true>, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, a, b, a2, b2, B, d, rows, D / 4);
    else hipLaunchKernelGGL(fusion_kernel<false>, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, a, b, a2, b2, B, d, rows, D / 4);
    SBL_LAUNCH_CHECK(who);
    return 0;
}
extern "C" int sbl_fusion_seg_fwd(const float* a, const float* b, float* a2, float* b2, int B, const int* seg_L, int nseg,
                                  int D, sbl_stream_t stream) {
    return fusion_common("sbl_fusion_fwd", a, b, a2, b2, B, seg_L, nseg, D, false, stream);
}
extern "C" int sbl_fusion_seg_bwd(const float* da2, const float* db2, float* da, float* db, int B, const int* seg_L, int nseg,
                                  int D, sbl_stream_t stream) {
    return fusion_common("sbl_fusion_bwd", da2, db2, da, db, B, seg_L, nseg, D, true, stream);
}
extern "C" int sbl_fusion_fwd(const float* a, const float* b, float* a2, float* b2, int B, int L, int D, sbl_stream_t stream) {
    return fusion_common("sbl_fusion_fwd", a, b, a2, b2, B, &L, 1, D, false, stream);
}
extern "C" int sbl_fusion_bwd(const float* da2, const float* db2, float* da, float* db, int B, int L, int D, sbl_stream_t stream) {
    return fusion_common("sbl_fusion_bwd", da2, db2, da, db, B, &L, 1, D, true, stream);
}

// ------------------------------------------------------------------ last position of every sequence: decoder.py:166-167
// out[s*B + b, :] = x[row_off[s] + b*L[s] + L[s]-1, :]   (the rows the two output heads read); bwd scatters back.
template <bool BWD>
__global__ __launch_bounds__(256) void gather_last_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                          SegDesc segs, int D4) {
    const long n4 = (long)segs.nseg * B * D4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const int sb = (int)(i / D4);
        const int s = sb / B, b = sb - s * B;
        const long row = segs.row_off[s] + (long)b * segs.L[s] + segs.L[s] - 1;
        if (!BWD) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[row * D4 + c];
        else reinterpret_cast<float4*>(dst)[row * D4 + c] = reinterpret_cast<const float4*>(src)[i];
    }
}
extern "C" int sbl_gather_last_fwd(const float* x, float* out, int B, const int* seg_L, int nseg, int D, sbl_stream_t stream) {
    SegDesc d;
    SBL_REQUIRE(sbl_make_segs(d, seg_L, nseg, B, 1, 1) > 0, "sbl_gather_last_fwd: bad segment list");
    SBL_REQUIRE(x && out && B > 0 && D > 0 && D % 4 == 0 && sbl_aligned16(x) && sbl_aligned16(out), "sbl_gather_last_fwd: bad args");
    hipLaunchKernelGGL(gather_last_kernel<false>, dim3(ew_grid((long)nseg * B * D / 4)), dim3(256), 0, (hipStream_t)stream, x, out, B, d, D / 4);
    SBL_LAUNCH_CHECK("sbl_gather_last_fwd");
    return 0;
}
// dx (rows x D) is zero-filled by the call, then the nseg*B gradient rows are scattered into it
extern "C" int sbl_gather_last_bwd(const float* dy, float* dx, int B, const int* seg_L, int nseg, int D, sbl_stream_t stream) {
    SegDesc d;
    const long rows = sbl_make_segs(d, seg_L, nseg, B, 1, 1);
    SBL_REQUIRE(rows > 0, "sbl_gather_last_bwd: bad segment list");
    SBL_REQUIRE(dy && dx && B > 0 && D > 0 && D % 4 == 0 && sbl_aligned16(dy) && sbl_aligned16(dx), "sbl_gather_last_bwd: bad args");
    SBL_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)rows * D, (hipStream_t)stream));
    hipLaunchKernelGGL(gather_last_kernel<true>, dim3(ew_grid((long)nseg * B * D / 4)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, d, D / 4);
    SBL_LAUNCH_CHECK("sbl_gather_last_bwd");
    return 0;
}

// ------------------------------------------------------------------ token feedback: decoder.py:173-186
// one wavefront per batch row; argmax returns the FIRST maximal index (torch.argmax tie-break on CPU)
__global__ __launch_bounds__(256) void argmax_select_kernel(const float* __restrict__ pred, long ldp,
                                                            const int64_t* __restrict__ gold, long ldg,
                                                            int64_t* __restrict__ ys, long ldy, int step, int use_argmax,
                                                            const int32_t* __restrict__ coins, int B, int V) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int own = coins ? coins[step] : use_argmax;
    if (!own) {
        if (lane == 0) ys[(long)b * ldy + step + 1] = gold[(long)b * ldg + step];
        return;
    }
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
        const float v = pred[(long)b * ldp + c];
        if (v > best) {
            best = v;
            bi = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) ys[(long)b * ldy + step + 1] = bi == 0x7fffffff ? 0 : bi;
}
extern "C" int sbl_argmax_select(const float* pred, long ldp, const int64_t* gold, long ldg, int64_t* ys, long ldy, int step,
                                 int use_argmax, const int32_t* coins_dev, int B, int V, sbl_stream_t stream) {
    SBL_REQUIRE(ys && B > 0 && V > 0 && ldp >= V && step >= 0 && step + 1 < ldy, "sbl_argmax_select: bad args");
    SBL_REQUIRE(pred || (!use_argmax && !coins_dev), "sbl_argmax_select: logits required unless the token is the gold one");
    SBL_REQUIRE(gold || (use_argmax && !coins_dev), "sbl_argmax_select: gold required unless always-argmax");
    SBL_REQUIRE(!gold || step < ldg, "sbl_argmax_select: step beyond gold width");
    hipLaunchKernelGGL(argmax_select_kernel, dim3(sbl_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, pred, ldp, gold, ldg,
                       ys, ldy, step, use_argmax, coins_dev, B, V);
    SBL_LAUNCH_CHECK("sbl_argmax_select");
    return 0;
}

// ------------------------------------------------------------------ stage head: embedding + PE + dropout, both directions
// decoder.py:116-120 for the l2r and r2l token buffers of one stage in ONE launch (blockIdx.y = direction): what
// sbl_embed_pe_seg_fwd + sbl_dropout did in four.  The mask is a function of (seed, offset_d, element index inside this
// stage's rows), exactly the indexing of dropout_kernel, so the backward regenerates it with sbl_dropout as before.
__global__ __launch_bounds__(256) void embed_pe_drop2_kernel(const int64_t* __restrict__ tok0, const int64_t* __restrict__ tok1, long ldt,
                                                             const float* __restrict__ emb, const float* __restrict__ pe,
                                                             float* __restrict__ out0, float* __restrict__ out1, int B, SegDesc segs,
                                                             long rows, int D4, int V, uint32_t thresh, float keep_scale,
                                                             const uint64_t* __restrict__ seed, uint64_t offset0, uint64_t offset1) {
    const int64_t* tok = blockIdx.y ? tok1 : tok0;
    float* out = blockIdx.y ? out1 : out0;
    const uint64_t offset = blockIdx.y ? offset1 : offset0;
    const uint64_t sd = thresh ? *seed : 0;
    const long n4 = rows * D4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const int r = (int)(i / D4);
        const int s = sbl_seg_of_row(segs, r, B);
        const int L = segs.L[s], rr = r - segs.row_off[s];
        const int b = rr / L, l = rr - b * L;
        long t = tok[(long)b * ldt + l];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);
        const float4 e = reinterpret_cast<const float4*>(emb)[t * D4 + c];
        const float4 p = reinterpret_cast<const float4*>(pe)[(long)l * D4 + c];
        float4 v = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
        if (thresh) {
            const uint64_t i0 = (uint64_t)i * 4;
            v.x = sbl_keep(sd, offset, i0 + 0, thresh) ? v.x * keep_scale : 0.f;
            v.y = sbl_keep(sd, offset, i0 + 1, thresh) ? v.y * keep_scale : 0.f;
            v.z = sbl_keep(sd, offset, i0 + 2, thresh) ? v.z * keep_scale : 0.f;
            v.w = sbl_keep(sd, offset, i0 + 3, thresh) ? v.w * keep_scale : 0.f;
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
extern "C" int sbl_embed_pe_drop2_fwd(const int64_t* tok0, const int64_t* tok1, long ldt, const float* emb, const float* pe,
                                      float* out0, float* out1, int B, const int* seg_L, int nseg, int D, int V, float drop_p,
                                      const uint64_t* seed, uint64_t offset0, uint64_t offset1, sbl_stream_t stream) {
    SegDesc d;
    const long rows = sbl_make_segs(d, seg_L, nseg, B, 1, 1);
    SBL_REQUIRE(rows > 0, "sbl_embed_pe_drop2_fwd: bad segment list");
    SBL_REQUIRE(tok0 && tok1 && emb && pe && out0 && out1 && B > 0 && D > 0 && D % 4 == 0 && V > 0, "sbl_embed_pe_drop2_fwd: bad args");
    SBL_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sbl_embed_pe_drop2_fwd: bad dropout args");
    for (int s = 0; s < nseg; ++s) SBL_REQUIRE(ldt >= seg_L[s], "sbl_embed_pe_drop2_fwd: token row shorter than prefix %d", seg_L[s]);
    SBL_REQUIRE(sbl_aligned16(emb) && sbl_aligned16(pe) && sbl_aligned16(out0) && sbl_aligned16(out1), "sbl_embed_pe_drop2_fwd: unaligned");
    hipLaunchKernelGGL(embed_pe_drop2_kernel, dim3(ew_grid(rows * D / 4), 2), dim3(256), 0, (hipStream_t)stream, tok0, tok1, ldt, emb, pe,
                       out0, out1, B, d, rows, D / 4, V, drop_p > 0.f ? sbl_drop_thresh(drop_p) : 0u, 1.f / (1.f - drop_p), seed, offset0,
                       offset1);
    SBL_LAUNCH_CHECK("sbl_embed_pe_drop2_fwd");
    return 0;
}

// ------------------------------------------------------------------ stage tail: last fusion + heads + token feedback
// After the last decoder layer of a stage the reference fuses the two directions (decoder.py:160-164), takes the LAST
// position of every sequence (:166-167), applies the two bias-free Linear(512, 58) heads and feeds the arg-max back
// (:173-186).  Only the last positions are ever read from that fusion: A'[L-1] = A[L-1] + B[0], B'[L-1] = 2 B[L-1] + A[0]
// (time flip along the sequence's own prefix).  The kernel forms that 512-vector per (direction, segment, batch row)
// (kept as `last`: the heads' weight gradient reads it), its 58 logits (fp32 FMA, wave-shuffle sums) and, for the
// stage's final step when the coin says "own arg-max", the next token (first maximal index, torch.argmax's CPU rule).
// One launch instead of fusion + 2 x (gather_last + GEMM + argmax_select).
// Workgroup = 4 rows x 64 class slots: wavefront w owns classes 16w .. 16w+15 for all four rows (a weight row is loaded once
// and used four times; all 32 of a wavefront's 16-byte weight loads are independent and issued up front), each lane forms
// 64 partial sums (row, class) over its 8 of the 512 columns, and a transposing butterfly (32+16+8+4+2+1 shuffles) leaves
// lane (r, cl) with the full dot product of row r and class 16w + cl.  (A first version looped over the 58 classes with one
// 6-shuffle wave sum each: 35 us per launch, the L2 round trip of every class exposed.)
__global__ __launch_bounds__(256) void decoder_tail_kernel(const float* __restrict__ yf0, const float* __restrict__ yf1,
                                                           const float* __restrict__ w0, const float* __restrict__ w1,
                                                           float* __restrict__ last0, float* __restrict__ last1,
                                                           float* __restrict__ pred0, float* __restrict__ pred1, long ldp,
                                                           int64_t* __restrict__ ys0, int64_t* __restrict__ ys1, long ldy, int step,
                                                           int write_tok, int B, SegDesc segs, int V) {
    __shared__ float s_logit[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int nrows = segs.nseg * B;
    const float kf = dir ? 2.f : 1.f;
    const float* ya = dir ? yf1 : yf0;      // this direction
    const float* yb = dir ? yf0 : yf1;      // the other one
    float x[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sb = min((int)blockIdx.x * 4 + r, nrows - 1);      // clamped: rows past the end are computed, never stored
        const int s = sb / B, b = sb - s * B;
        const int L = segs.L[s];
        const long r0 = segs.row_off[s] + (long)b * L, rl = r0 + L - 1;
        const float* own = ya + rl * 512;      // last position
        const float* oth = yb + r0 * 512;      // the other direction's first (= time-flipped last) position
        const float4 a0 = *reinterpret_cast<const float4*>(own + lane * 4), a1 = *reinterpret_cast<const float4*>(own + 256 + lane * 4);
        const float4 c0 = *reinterpret_cast<const float4*>(oth + lane * 4), c1 = *reinterpret_cast<const float4*>(oth + 256 + lane * 4);
        x[r][0] = kf * a0.x + c0.x; x[r][1] = kf * a0.y + c0.y; x[r][2] = kf * a0.z + c0.z; x[r][3] = kf * a0.w + c0.w;
        x[r][4] = kf * a1.x + c1.x; x[r][5] = kf * a1.y + c1.y; x[r][6] = kf * a1.z + c1.z; x[r][7] = kf * a1.w + c1.w;
    }
    {   // wavefront w stores row w of `last`
        const int sb = (int)blockIdx.x * 4 + wave;
        if (sb < nrows) {
            float* lo = (dir ? last1 : last0) + (long)sb * 512;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r == wave) {
                    *reinterpret_cast<float4*>(lo + lane * 4) = make_float4(x[r][0], x[r][1], x[r][2], x[r][3]);
                    *reinterpret_cast<float4*>(lo + 256 + lane * 4) = make_float4(x[r][4], x[r][5], x[r][6], x[r][7]);
                }
        }
    }
    const float* w = dir ? w1 : w0;
    float v[64];      // v[r * 16 + cl]: partial dot product of row r and class 16 * wave + cl over this lane's 8 columns
    float4 u0[16], u1[16];
#pragma unroll
    for (int cl = 0; cl < 16; ++cl) {
        const int c = min(wave * 16 + cl, V - 1);
        u0[cl] = *reinterpret_cast<const float4*>(w + (long)c * 512 + lane * 4);
        u1[cl] = *reinterpret_cast<const float4*>(w + (long)c * 512 + 256 + lane * 4);
    }
#pragma unroll
    for (int cl = 0; cl < 16; ++cl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float acc = x[r][0] * u0[cl].x;
            acc = fmaf(x[r][1], u0[cl].y, acc); acc = fmaf(x[r][2], u0[cl].z, acc); acc = fmaf(x[r][3], u0[cl].w, acc);
            acc = fmaf(x[r][4], u1[cl].x, acc); acc = fmaf(x[r][5], u1[cl].y, acc); acc = fmaf(x[r][6], u1[cl].z, acc);
            acc = fmaf(x[r][7], u1[cl].w, acc);
            v[r * 16 + cl] = acc;
        }
    // transposing butterfly: after the step with offset o, a lane keeps the half of its values whose index has bit o equal to
    // its own lane bit o; lane l ends with the sum over all lanes of v[l]
#define SBL_TSTEP(N, O)                                                       \
    _Pragma("unroll") for (int j = 0; j < (N) / 2; ++j) {                     \
        const bool hi = (lane & (O)) != 0;                                    \
        const float keep = hi ? v[j + (N) / 2] : v[j];                        \
        const float send = hi ? v[j] : v[j + (N) / 2];                        \
        v[j] = keep + __shfl_xor(send, (O), 64);                              \
    }
    SBL_TSTEP(64, 32) SBL_TSTEP(32, 16) SBL_TSTEP(16, 8) SBL_TSTEP(8, 4) SBL_TSTEP(4, 2) SBL_TSTEP(2, 1)
#undef SBL_TSTEP
    {
        const int r = lane >> 4, c = wave * 16 + (lane & 15);
        const int sb = (int)blockIdx.x * 4 + r;
        s_logit[r][c] = c < V ? v[0] : -INFINITY;
        if (sb < nrows && c < V) (dir ? pred1 : pred0)[(long)sb * ldp + c] = v[0];
    }
    if (!write_tok) return;      // (workgroup-uniform)
    __syncthreads();
    const int sb = (int)blockIdx.x * 4 + wave;
    if (sb >= nrows || sb / B != segs.nseg - 1) return;      // token feedback: the stage's final step only
    float best = s_logit[wave][lane];
    int bi = lane < V ? lane : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) (dir ? ys1 : ys0)[(long)(sb - (segs.nseg - 1) * B) * ldy + step + 1] = bi == 0x7fffffff ? 0 : bi;
}
extern "C" int sbl_decoder_tail_fwd(const float* yf0, const float* yf1, const float* w0, const float* w1, float* last0, float* last1,
                                    float* pred0, float* pred1, long ldp, int64_t* ys0, int64_t* ys1, long ldy, int step,
                                    int write_tok, int B, const int* seg_L, int nseg, int D, int V, sbl_stream_t stream) {
    SegDesc d;
    SBL_REQUIRE(sbl_make_segs(d, seg_L, nseg, B, 1, 1) > 0, "sbl_decoder_tail_fwd: bad segment list");
    SBL_REQUIRE(D == 512 && V >= 1 && V <= 64 && ldp >= V && B > 0, "sbl_decoder_tail_fwd: built for d_model = 512 and at most 64 classes (D=%d V=%d)", D, V);
    SBL_REQUIRE(yf0 && yf1 && w0 && w1 && last0 && last1 && pred0 && pred1, "sbl_decoder_tail_fwd: null operand");
    SBL_REQUIRE(!write_tok || (ys0 && ys1 && step >= 0 && step + 1 < ldy), "sbl_decoder_tail_fwd: token feedback needs ys buffers and step + 1 < ldy");
    SBL_REQUIRE(sbl_aligned16(yf0) && sbl_aligned16(yf1) && sbl_aligned16(w0) && sbl_aligned16(w1) && sbl_aligned16(last0) && sbl_aligned16(last1),
                "sbl_decoder_tail_fwd: unaligned");
    hipLaunchKernelGGL(decoder_tail_kernel, dim3(sbl_cdiv((long)nseg * B, 4), 2), dim3(256), 0, (hipStream_t)stream, yf0, yf1, w0, w1, last0,
                       last1, pred0, pred1, ldp, ys0, ys1, ldy, step, write_tok, B, d, V);
    SBL_LAUNCH_CHECK("sbl_decoder_tail_fwd");
    return 0;
}

// ------------------------------------------------------------------ Decoder.preprocess, decoder.py:62-77
// One thread per target row: strip IGNORE_ID keeping the order, <sos> in front of the input form, <eos> padding to maxlen in
// both forms.  set = 0 / 1: the l2r / r2l targets of one step in one launch (torch's argsort + gather + where + fills were
// ~20 launches per direction).
__global__ void decoder_preprocess_kernel(const int64_t* __restrict__ p0, const int64_t* __restrict__ p1, int64_t* __restrict__ in0,
                                          int64_t* __restrict__ out0, int64_t* __restrict__ in1, int64_t* __restrict__ out1, int N,
                                          int To, int maxlen, int64_t sos, int64_t eos, int64_t ignore) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t* p = (blockIdx.y ? p1 : p0) + (long)i * To;
    int64_t* yi = (blockIdx.y ? in1 : in0) + (long)i * maxlen;
    int64_t* yo = (blockIdx.y ? out1 : out0) + (long)i * maxlen;
    for (int t = 0; t < maxlen; ++t) {
        yi[t] = eos;
        yo[t] = eos;
    }
    yi[0] = sos;
    int c = 0;
    for (int t = 0; t < To; ++t) {
        const int64_t v = p[t];
        if (v != ignore) {
            if (c + 1 < maxlen) yi[1 + c] = v;
            if (c < maxlen) yo[c] = v;
            ++c;
        }
    }
}
extern "C" int sbl_decoder_preprocess(const int64_t* padded0, const int64_t* padded1, int64_t* ys_in0, int64_t* ys_out0,
                                      int64_t* ys_in1, int64_t* ys_out1, int N, int To, int maxlen, int64_t sos, int64_t eos,
                                      int64_t ignore, sbl_stream_t stream) {
    SBL_REQUIRE(padded0 && ys_in0 && ys_out0 && N > 0 && To > 0 && maxlen > 1, "sbl_decoder_preprocess: bad args N=%d To=%d maxlen=%d", N, To, maxlen);
    SBL_REQUIRE(!padded1 || (ys_in1 && ys_out1), "sbl_decoder_preprocess: second target set without outputs");
    hipLaunchKernelGGL(decoder_preprocess_kernel, dim3(sbl_cdiv(N, 64), padded1 ? 2 : 1), dim3(64), 0, (hipStream_t)stream, padded0,
                       padded1, ys_in0, ys_out0, ys_in1, ys_out1, N, To, maxlen, sos, eos, ignore);
    SBL_LAUNCH_CHECK("sbl_decoder_preprocess");
    return 0;
}

// ------------------------------------------------------------------ label-smoothed CE: loss.py:27-52
// one wavefront per row (C <= 64*4).  q = onehot*(1-eps) + (1-onehot)*eps/C (rows do not sum to 1: kept).
// eps == 0 reduces to plain cross entropy with ignore_index (loss.py:48-50).
__global__ __launch_bounds__(256) void smoothed_ce_fwd_kernel(const float* __restrict__ pred, const int64_t* __restrict__ gold,
                                                              float* __restrict__ out3, int R, int C, float eps, int ignore_id) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const long g = gold[r];
    if (g == ignore_id) return;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = pred[(long)r * C + c];
        if (v > mx) {
            mx = v;
            am = c;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oi < am)) {
            mx = ov;
            am = oi;
        }
    }
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(pred[(long)r * C + c] - mx);
    const float lse = __logf(wave_sum(se)) + mx;
    float loss = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float qv = (c == g) ? (1.f - eps) : eps / (float)C;
        loss -= qv * (pred[(long)r * C + c] - lse);
    }
    loss = wave_sum(loss);
    if (lane == 0) {
        atomicAdd(out3 + 0, loss);
        atomicAdd(out3 + 1, 1.f);
        if (am == g) atomicAdd(out3 + 2, 1.f);
    }
}
// d loss_mean / d pred[r][c] = gscale/n_valid * (sum(q) * softmax - q) on valid rows, 0 elsewhere
__global__ __launch_bounds__(256) void smoothed_ce_bwd_kernel(const float* __restrict__ pred, const int64_t* __restrict__ gold,
                                                              const float* __restrict__ out3, const float* __restrict__ gscale,
                                                              float* __restrict__ dpred, int R, int C, float eps, int ignore_id) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const long g = gold[r];
    if (g == ignore_id) {
        for (int c = lane; c < C; c += 64) dpred[(long)r * C + c] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, pred[(long)r * C + c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(pred[(long)r * C + c] - mx);
    se = wave_sum(se);
    const float scale = gscale[0] / out3[1];
    const float qsum = (1.f - eps) + (float)(C - 1) * eps / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float sm = __expf(pred[(long)r * C + c] - mx) / se;
        const float qv = (c == g) ? (1.f - eps) : eps / (float)C;
        dpred[(long)r * C + c] = scale * (qsum * sm - qv);
    }
}
extern "C" int sbl_smoothed_ce_fwd(const float* pred, const int64_t* gold, float* out3, int R, int C, float eps, int ignore_id,
                                   sbl_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SBL_REQUIRE(pred && gold && out3 && R > 0 && C > 0 && eps >= 0.f && eps < 1.f, "sbl_smoothed_ce_fwd: bad args");
    SBL_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(float), s));
    hipLaunchKernelGGL(smoothed_ce_fwd_kernel, dim3(sbl_cdiv(R, 4)), dim3(256), 0, s, pred, gold, out3, R, C, eps, ignore_id);
    SBL_LAUNCH_CHECK("sbl_smoothed_ce_fwd");
    return 0;
}
extern "C" int sbl_smoothed_ce_bwd(const float* pred, const int64_t* gold, const float* out3, const float* gscale,
                                   float* dpred, int R, int C, float eps, int ignore_id, sbl_stream_t stream) {
    SBL_REQUIRE(pred && gold && out3 && gscale && dpred && R > 0 && C > 0, "sbl_smoothed_ce_bwd: bad args");
    hipLaunchKernelGGL(smoothed_ce_bwd_kernel, dim3(sbl_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, pred, gold, out3,
                       gscale, dpred, R, C, eps, ignore_id);
    SBL_LAUNCH_CHECK("sbl_smoothed_ce_bwd");
    return 0;
}

// ------------------------------------------------------------------ fused Adam over a flat buffer
// torch.optim.Adam semantics (SBL/train.py:75): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float b1, float b2, float eps,
                                                   float step_size, float inv_sqrt_bc2, float gscale) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}
extern "C" int sbl_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, sbl_stream_t stream) {
    SBL_REQUIRE(p && g && m && v && n > 0 && step >= 1, "sbl_adam_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
    SBL_LAUNCH_CHECK("sbl_adam_step");
    return 0;
}

// ------------------------------------------------------------------ y[m,:] = x[m,:] * s[m]
// the `*= non_pad_mask` of encoder.py:86,89 / decoder.py:399,403,406 for ragged input_lengths
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       float* __restrict__ y, long n4, int D4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float f = s[i / D4];
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<float4*>(y)[i] = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
    }
}
extern "C" int sbl_rowscale(const float* x, const float* s, float* y, long M, int D, sbl_stream_t stream) {
    SBL_REQUIRE(x && s && y && M > 0 && D > 0 && D % 4 == 0 && sbl_aligned16(x) && sbl_aligned16(y), "sbl_rowscale: bad args");
    const long n4 = M * (D / 4);
    hipLaunchKernelGGL(rowscale_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, s, y, n4, D / 4);
    SBL_LAUNCH_CHECK("sbl_rowscale");
    return 0;
}

// ------------------------------------------------------------------ device input pipeline
// uint8 grayscale frames -> the fp32 clips the stem reads: /255, ColorNormalize, crop, per-clip horizontal flip
// (cvtransforms.py:7-48), frame removal + zero padding to Tout frames (data_gen.py:104-108, 290-296) expressed as a
// source-frame map (-1 = zero frame).  The 256 possible pixel values go through a LUT computed in double on the
// host, so the result is bit-identical to the reference's float64 numpy arithmetic cast to float32.
__global__ __launch_bounds__(256) void preprocess_clips_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                               const float* __restrict__ lut, const int* __restrict__ y1,
                                                               const int* __restrict__ x1, const int* __restrict__ flip,
                                                               const int* __restrict__ src_frame, int N, int Tin, int Hin,
                                                               int Win, int Tout, int Hc, int Wc) {
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const long total = (long)N * Tout * Hc * Wc;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % Wc);
        long r = i / Wc;
        const int y = (int)(r % Hc);
        r /= Hc;
        const int t = (int)(r % Tout);
        const int n = (int)(r / Tout);
        const int st = src_frame[n * Tout + t];
        float v = 0.f;
        if (st >= 0) {
            const int sx = x1[n] + (flip[n] ? (Wc - 1 - x) : x);
            v = s_lut[in[(((long)n * Tin + st) * Hin + (y1[n] + y)) * Win + sx]];
        }
        out[i] = v;
    }
}
extern "C" int sbl_preprocess_clips(const uint8_t* in, float* out, const float* lut256, const int* y1, const int* x1,
                                    const int* flip, const int* src_frame, int N, int Tin, int Hin, int Win, int Tout, int Hc,
                                    int Wc, sbl_stream_t stream) {
    SBL_REQUIRE(in && out && lut256 && y1 && x1 && flip && src_frame, "sbl_preprocess_clips: null pointer");
    SBL_REQUIRE(N > 0 && Tin > 0 && Tout > 0 && Hc > 0 && Wc > 0 && Hc <= Hin && Wc <= Win, "sbl_preprocess_clips: bad dims");
    const long total = (long)N * Tout * Hc * Wc;
    hipLaunchKernelGGL(preprocess_clips_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, in, out, lut256, y1, x1,
                       flip, src_frame, N, Tin, Hin, Win, Tout, Hc, Wc);
    SBL_LAUNCH_CHECK("sbl_preprocess_clips");
    return 0;
}
