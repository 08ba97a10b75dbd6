// Operand loaders of the tile engine (mfma_gemm.h): how one workgroup fetches a 16-deep K slab of its A or B operand.
//
// A loader presents element (r, k) of a GEMM operand (r = row of the tile, k = contraction index).  Every thread owns a
// fixed set of float4s per slab (Regs), fetched by load() and written to the LDS image by store() (fp32 body) or by
// bf_store() (bf16 bodies, bf16_tile.h).  Two families, told apart by kKC:
//   k-contiguous ("KC"): thread (tid>>2) + 64*ps is a row, (tid&3)*4 its 4 consecutive k   -> Regs.v[BR/64]
//   m-contiguous ("MC"): thread tid/TPR (+ RPP*ps) is a k row, (tid%TPR)*4 its 4 consecutive r -> Regs.v[16/RPP]
//
// All fetches are raw buffer loads (buffer_load_dwordx4 ... offen) off a wave-uniform descriptor of the operand: an
// element that does not exist (row beyond M, k beyond K, a convolution tap in the zero padding) gets the offset SBL_OOB,
// which the hardware range check answers with zeros without touching memory.  No branch and no EXEC masking around
// a load, so the compiler keeps counted s_waitcnt vmcnt(N) waits (a load inside a divergent branch forces vmcnt(0),
// which would drain the two-steps-ahead prefetch of the bf16 bodies).  The descriptor covers 2 GiB from the operand's
// base (offsets are 32-bit): every operand must span less than that (checked on the host, sbl_fits_u32).
#pragma once
#include "sbl_common.h"

#define SBL_BK 16
#define SBL_BUF_BYTES 0x80000000u      // num_records of every operand descriptor
#define SBL_OOB 0xC0000000u            // an offset beyond it for every access width (no 32-bit wrap when the size is added)

typedef __amdgpu_buffer_rsrc_t sbl_rsrc;
__device__ __forceinline__ sbl_rsrc sbl_make_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)SBL_BUF_BYTES, 0x00020000);
}
__device__ __forceinline__ float4 sbl_ld4(sbl_rsrc r, unsigned off) {
    // (cast the whole vector: __builtin_bit_cast(float, v[i]) on the elements of the returned vector is miscompiled by
    // ROCm 7.2's clang into one dword load replicated four times)
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float sbl_ld1(sbl_rsrc r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
static inline bool sbl_fits_u32(long elems) { return elems >= 0 && elems * 4 <= (long)SBL_BUF_BYTES; }

// ---- shared LDS stores of the fp32 body (image [k][BR + 4], k-major)
template <int BR, class Regs>
__device__ __forceinline__ void sbl_store_kc(float (*lds)[BR + 4], const Regs& r, int tid) {
    const int kq = (tid & 3) * 4;
#pragma unroll
    for (int ps = 0; ps < BR / 64; ++ps) {
        const int row = ps * 64 + (tid >> 2);
        lds[kq + 0][row] = r.v[ps].x;
        lds[kq + 1][row] = r.v[ps].y;
        lds[kq + 2][row] = r.v[ps].z;
        lds[kq + 3][row] = r.v[ps].w;
    }
}
template <int BR, class Regs>
__device__ __forceinline__ void sbl_store_mc(float (*lds)[BR + 4], const Regs& r, int tid) {
    constexpr int TPR = BR / 4, RPP = 256 / TPR, NP = SBL_BK / RPP;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) *reinterpret_cast<float4*>(&lds[tid / TPR + ps * RPP][(tid % TPR) * 4]) = r.v[ps];
}
template <class Regs, int N>
__device__ __forceinline__ void sbl_accum(const Regs& r, float4& cs) {
#pragma unroll
    for (int ps = 0; ps < N; ++ps) {
        cs.x += r.v[ps].x; cs.y += r.v[ps].y; cs.z += r.v[ps].z; cs.w += r.v[ps].w;
    }
}

// ------------------------------------------------------------------ dense
// Dense, k-contiguous: element (r,k) at p[r*ld + k].  Used for X[M,K] and W[N,K] of Linear.
template <int BR, bool VEC>
struct DenseKC {
    static constexpr bool kColSum = false;
    static constexpr bool kKC = true;
    static constexpr bool kDenseOperand = true;      // plain nn.Linear operand (the dense launches may split K over wave groups)
    const float* p;
    long ld;
    int rows;
    struct State {
        sbl_rsrc rs;
        unsigned ro[BR / 64];     // byte offset of (row, kq), SBL_OOB for rows beyond the operand
        int kq;
    };
    struct Regs {
        float4 v[BR / 64];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        s.kq = (tid & 3) * 4;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const int r = r0 + ps * 64 + (tid >> 2);
            s.ro[ps] = r < rows ? (unsigned)(((long)r * ld + s.kq) * 4) : SBL_OOB;
        }
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
        const int k = k0 + s.kq;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const bool ok = s.ro[ps] != SBL_OOB;
            const unsigned o = s.ro[ps] + (unsigned)k0 * 4;
            if (VEC) {
                r.v[ps] = sbl_ld4(s.rs, ok && k < kend ? o : SBL_OOB);      // K % 4 == 0 on this path
            } else {
                r.v[ps].x = sbl_ld1(s.rs, ok && k + 0 < kend ? o + 0 : SBL_OOB);
                r.v[ps].y = sbl_ld1(s.rs, ok && k + 1 < kend ? o + 4 : SBL_OOB);
                r.v[ps].z = sbl_ld1(s.rs, ok && k + 2 < kend ? o + 8 : SBL_OOB);
                r.v[ps].w = sbl_ld1(s.rs, ok && k + 3 < kend ? o + 12 : SBL_OOB);
            }
        }
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_kc<BR>(lds, r, tid); }
};

// Dense k-contiguous rows whose k axis is a SUBSET of C-wide blocks of the stored row: GEMM-k block t (k in
// [t*C, (t+1)*C)) lives at stored block lin[t].  Used for the weight operand of the parity-class input gradients
// (the class's taps out of the [Cin][KH*KW][Cout] rows).  C % 16 == 0, so a BK slice never straddles a block.
template <int BR>
struct DenseKCTaps {
    static constexpr bool kColSum = false;
    static constexpr bool kKC = true;
    static constexpr bool kDenseOperand = false;
    const float* p;
    long ld;
    int rows;
    int C;
    int lin[4];
    struct State {
        sbl_rsrc rs;
        unsigned ro[BR / 64];
        int kq;
    };
    struct Regs {
        float4 v[BR / 64];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        s.kq = (tid & 3) * 4;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const int r = r0 + ps * 64 + (tid >> 2);
            s.ro[ps] = r < rows ? (unsigned)(((long)r * ld + s.kq) * 4) : SBL_OOB;
        }
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
        const int t = k0 / C;              // block-uniform
        int l = lin[0];
#pragma unroll
        for (int u = 1; u < 4; ++u)
            if (u == t) l = lin[u];
        const unsigned kb = (unsigned)(l * C + (k0 - t * C)) * 4;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps)
            r.v[ps] = sbl_ld4(s.rs, s.ro[ps] != SBL_OOB && k0 < kend ? s.ro[ps] + kb : SBL_OOB);
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_kc<BR>(lds, r, tid); }
};

// Same, with the block list packed 4 bits per entry (up to 9 blocks) and set per workgroup: the weight operand of the
// position-major convolutions below, whose tiles contract only over the taps that can be in bounds for their pixels.
template <int BR>
struct DenseKCTapList {
    static constexpr bool kColSum = false;
    static constexpr bool kKC = true;
    static constexpr bool kDenseOperand = false;
    const float* p;
    long ld;
    int rows;
    int C;
    unsigned long long taps;
    struct State {
        sbl_rsrc rs;
        unsigned ro[BR / 64];
        int kq;
    };
    struct Regs {
        float4 v[BR / 64];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        s.kq = (tid & 3) * 4;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const int r = r0 + ps * 64 + (tid >> 2);
            s.ro[ps] = r < rows ? (unsigned)(((long)r * ld + s.kq) * 4) : SBL_OOB;
        }
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
        const int t = k0 / C;              // block-uniform
        const int l = (int)((taps >> (4 * (t & 15))) & 15ull);
        const unsigned kb = (unsigned)(l * C + (k0 - t * C)) * 4;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps)
            r.v[ps] = sbl_ld4(s.rs, s.ro[ps] != SBL_OOB && k0 < kend ? s.ro[ps] + kb : SBL_OOB);
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_kc<BR>(lds, r, tid); }
};

// Dense, m-contiguous: element (r,k) at p[k*ld + r].  Used for dY^T / X in weight-gradient
// GEMMs and for W[K,N] in input-gradient GEMMs.  VEC: 16-byte aligned rows and rows % 4 == 0.
template <int BR, bool VEC>
struct DenseMC {
    static constexpr bool kColSum = true;
    static constexpr bool kKC = false;
    static constexpr bool kDenseOperand = true;
    const float* p;
    long ld;
    int rows;
    static constexpr int TPR = BR / 4;           // threads per k-row
    static constexpr int RPP = 256 / TPR;        // k-rows per pass
    static constexpr int NP = SBL_BK / RPP;      // passes
    struct State {
        sbl_rsrc rs;
        int c, kr;
    };
    struct Regs {
        float4 v[NP];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        s.c = r0 + (tid % TPR) * 4;
        s.kr = tid / TPR;
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int k = k0 + s.kr + ps * RPP;
            const unsigned o = (unsigned)(((long)k * ld + s.c) * 4);
            const bool kin = k < kend;
            if constexpr (VEC) {       // rows % 4 == 0 (host-checked): a lane's four elements exist together
                r.v[ps] = sbl_ld4(s.rs, kin && s.c < rows ? o : SBL_OOB);
            } else {                   // ragged right edge / unaligned rows: per-element range checks
                r.v[ps].x = sbl_ld1(s.rs, kin && s.c + 0 < rows ? o + 0 : SBL_OOB);
                r.v[ps].y = sbl_ld1(s.rs, kin && s.c + 1 < rows ? o + 4 : SBL_OOB);
                r.v[ps].z = sbl_ld1(s.rs, kin && s.c + 2 < rows ? o + 8 : SBL_OOB);
                r.v[ps].w = sbl_ld1(s.rs, kin && s.c + 3 < rows ? o + 12 : SBL_OOB);
            }
        }
    }
    __device__ __forceinline__ void accum(const Regs& r, float4& cs) const { sbl_accum<Regs, NP>(r, cs); }   // column sums over k
    __device__ __forceinline__ int col(const State& s) const { return s.c; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_mc<BR>(lds, r, tid); }
};

// Segmented m-contiguous operand: the GEMM-k axis is the concatenation of up to 16 row blocks that live in
// different tensors (the per-stage activations / gradients of one decoder layer): element (r, k) = p[s][(k -
// kcum[s])*ld + r] with s the segment containing k.  Lets one weight-gradient GEMM contract over all decoder
// stages of a step (K ~ 4352 rows) instead of one skinny GEMM per stage.  (Plain global loads: the segment base is
// per lane when the segments are not slab aligned, so there is no wave-uniform descriptor.)
#define SBL_MAX_KSEG 16
// ALIGNED: every segment length is a multiple of SBL_BK, so a BK-deep slice lies in one segment and the segment
// lookup is workgroup-uniform (scalar unit) instead of 15 compare/select pairs per lane per load.
template <int BR, bool ALIGNED = false>
struct SegMC {
    static constexpr bool kColSum = true;
    static constexpr bool kKC = false;
    static constexpr bool kDenseOperand = false;
    const float* p[SBL_MAX_KSEG];
    int kcum[SBL_MAX_KSEG + 1];
    int nseg;
    long ld;
    int rows;
    static constexpr int TPR = BR / 4;
    static constexpr int RPP = 256 / TPR;
    static constexpr int NP = SBL_BK / RPP;
    struct State {
        int c, kr;
    };
    struct Regs {
        float4 v[NP];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.c = r0 + (tid % TPR) * 4;
        s.kr = tid / TPR;
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int k = k0 + s.kr + ps * RPP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) {
                const int kl = ALIGNED ? k0 : k;      // k0 is uniform: the lookup below then runs on the scalar unit
                int sg = 0;
#pragma unroll
                for (int t = 1; t < SBL_MAX_KSEG; ++t)
                    if (t < nseg && kl >= kcum[t]) sg = t;
                const float* q = p[sg] + (long)(k - kcum[sg]) * ld + s.c;
                if (s.c + 3 < rows) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (s.c + 0 < rows) v.x = q[0];
                    if (s.c + 1 < rows) v.y = q[1];
                    if (s.c + 2 < rows) v.z = q[2];
                    if (s.c + 3 < rows) v.w = q[3];
                }
            }
            r.v[ps] = v;
        }
    }
    __device__ __forceinline__ void accum(const Regs& r, float4& cs) const { sbl_accum<Regs, NP>(r, cs); }
    __device__ __forceinline__ int col(const State& s) const { return s.c; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_mc<BR>(lds, r, tid); }
};

// ------------------------------------------------------------------ convolutions (NHWC implicit GEMM)
// One geometry struct for fwd / dgrad / wgrad gathers.
//   "out" grid (OH,OW) indexes GEMM rows; "src" tensor (SH,SW,C) is what gets gathered.
//   fwd / wgrad:  src = x (H,W,Cin),  out = y grid (Ho,Wo):   ih = oh*stride - pad + kh
//   dgrad:        src = dy (Ho,Wo,Cout), out = x grid (H,W):  t = oh + pad - kh, valid iff t%stride==0,
//                                                              ih = t/stride < SH
// n / d for n < 2^31 by one multiply-high and a shift (Granlund-Montgomery with 31-bit numerators: m = ceil(2^(31+l) / d),
// l = ceil(log2 d), fits 32 bits); the per-slab index arithmetic of the gathers would otherwise spend ~20 instructions
// per division, which at one or two bf16 MFMAs per slab is what bounds the kernel.
struct FastDiv {
    unsigned m;
    int s;          // < 0: d == 1
};
static inline FastDiv sbl_fastdiv(unsigned d) {
    FastDiv f{0u, -1};
    if (d <= 1) return f;
    int l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
    f.s = l - 1;
    return f;
}
__device__ __forceinline__ unsigned sbl_div(unsigned n, const FastDiv& f) {
    const unsigned q = __umulhi(n, f.m) >> (f.s & 31);      // always computed: a select, not a branch, in the K loops
    return f.s < 0 ? n : q;
}

struct ConvGeom {
    int NIMG, OH, OW;   // GEMM-row grid
    int SH, SW, C;      // gathered tensor (NHWC)
    int KH, KW, stride, pad;
    // stride-2 input-gradient parity classes (cls != 0): the GEMM rows are the input pixels (2a + ph, 2b + pw) only,
    // OH/OW are that sub-grid's dims, and GEMM-k runs over the ntaps <= 4 taps (tkh[t], tkw[t]) that can reach
    // such a pixel (kh = ph + pad mod 2, kw likewise) instead of over all KH*KW taps, 3/4 of which would gather zeros.
    int cls, ph, pw, ntaps;
    int tkh[4], tkw[4];
    FastDiv fdC, fdKW, fdHW, fdOW, fdNIMG;      // filled by sbl_geom_finish (host)
};
static inline void sbl_geom_finish(ConvGeom& g) {
    g.fdC = sbl_fastdiv((unsigned)g.C);
    g.fdKW = sbl_fastdiv((unsigned)g.KW);
    g.fdHW = sbl_fastdiv((unsigned)(g.OH * g.OW));
    g.fdOW = sbl_fastdiv((unsigned)g.OW);
    g.fdNIMG = sbl_fastdiv((unsigned)g.NIMG);
}
// Source pixel of GEMM-row pixel (oh, ow) under tap t is (oh*sA + dh, ow*sA + dw): affine in every case the engine runs
// (forward any stride; input gradient stride 1; input gradient stride 2 by parity class, where (oh, ow) index the class
// sub-grid).  t indexes the KH*KW taps, or the class's tap list.
// MAYCLS = false: the caller never runs parity classes (the class tap arrays of a kernel-argument copy that the kernel
// also writes to - the position-major tap list - would otherwise be kept in scratch memory).
template <bool DGRAD, bool MAYCLS = DGRAD>
__device__ __forceinline__ void conv_tap_delta(const ConvGeom& g, int t, int& dh, int& dw) {
    int kh, kw;
    if (MAYCLS && g.cls) {
        kh = g.tkh[0]; kw = g.tkw[0];
#pragma unroll
        for (int u = 1; u < 4; ++u)
            if (u == t) { kh = g.tkh[u]; kw = g.tkw[u]; }
    } else {
        kh = (int)sbl_div((unsigned)t, g.fdKW);
        kw = t - kh * g.KW;
    }
    if (!DGRAD) { dh = kh - g.pad; dw = kw - g.pad; }
    else if (MAYCLS && g.cls) { dh = (g.ph + g.pad - kh) >> 1; dw = (g.pw + g.pad - kw) >> 1; }
    else { dh = g.pad - kh; dw = g.pad - kw; }
}
template <bool DGRAD>
__device__ __forceinline__ int conv_row_scale(const ConvGeom& g) { return DGRAD ? 1 : g.stride; }
// bit t of the result: tap t of pixel (oh, ow) lies inside the source map (ntap = number of taps to test)
template <bool DGRAD, bool MAYCLS = DGRAD>
__device__ __forceinline__ unsigned conv_tap_mask(const ConvGeom& g, int oh, int ow, int ntap) {
    const int sA = conv_row_scale<DGRAD>(g);
    unsigned m = 0;
    for (int t = 0; t < ntap; ++t) {
        int dh, dw;
        conv_tap_delta<DGRAD, MAYCLS>(g, t, dh, dw);
        const int ih = oh * sA + dh, iw = ow * sA + dw;
        if ((unsigned)ih < (unsigned)g.SH && (unsigned)iw < (unsigned)g.SW) m |= 1u << t;
    }
    return m;
}

// im2col rows, k-contiguous: row = output pixel, k = (tap, c) with c fastest.  C % 16 == 0
// so one BK slice never straddles a tap and each lane's float4 stays inside one pixel.  Per row the thread keeps the
// byte offset of its pixel's origin and the bit mask of its in-bounds taps; a slab then costs one add, one bit test and
// one select per load (the tap's offset is workgroup-uniform: scalar unit).
template <int BR, bool DGRAD>
struct ConvGatherKC {
    static constexpr bool kColSum = false;
    static constexpr bool kKC = true;
    static constexpr bool kDenseOperand = false;
    const float* p;
    ConvGeom g;
    int rows;   // NIMG*OH*OW
    struct State {
        sbl_rsrc rs;
        int base[BR / 64];          // byte offset of (img, oh*sA, ow*sA, kq); may be "negative" before the tap is added
        unsigned mask[BR / 64];     // in-bounds taps (0 for rows beyond the operand)
    };
    struct Regs {
        float4 v[BR / 64];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        const int sA = conv_row_scale<DGRAD>(g);
        const int ntap = (DGRAD && g.cls) ? g.ntaps : g.KH * g.KW;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const int r = r0 + ps * 64 + (tid >> 2);
            const bool ok = r < rows;
            const unsigned rr = ok ? (unsigned)r : 0u;
            const int img = (int)sbl_div(rr, g.fdHW);
            const int rem = (int)rr - img * (g.OH * g.OW);
            const int oh = (int)sbl_div((unsigned)rem, g.fdOW), ow = rem - oh * g.OW;
            s.base[ps] = (((img * g.SH + oh * sA) * g.SW + ow * sA) * g.C + (tid & 3) * 4) * 4;
            s.mask[ps] = ok ? conv_tap_mask<DGRAD>(g, oh, ow, ntap) : 0u;
        }
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
        const int tap = (int)sbl_div((unsigned)k0, g.fdC);          // workgroup-uniform
        int dh, dw;
        conv_tap_delta<DGRAD>(g, tap, dh, dw);
        const int delta = ((dh * g.SW + dw) * g.C + (k0 - tap * g.C)) * 4;
        const unsigned bit = k0 < kend ? (1u << (tap & 31)) : 0u;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) r.v[ps] = sbl_ld4(s.rs, (s.mask[ps] & bit) ? (unsigned)(s.base[ps] + delta) : SBL_OOB);
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_kc<BR>(lds, r, tid); }
};

// im2col columns, m-contiguous (weight gradient B operand): GEMM-k = output pixel,
// GEMM-row r = (kh,kw,c) with c fastest; a lane's 4 consecutive r share one tap (C % 4 == 0).
template <int BR>
struct ConvGatherMC {
    static constexpr bool kColSum = false;
    static constexpr bool kKC = false;
    static constexpr bool kDenseOperand = false;
    const float* p;
    ConvGeom g;
    int rows;   // KH*KW*C
    static constexpr int TPR = BR / 4;
    static constexpr int RPP = 256 / TPR;
    static constexpr int NP = SBL_BK / RPP;
    struct State {
        sbl_rsrc rs;
        int c, dh, dw, kr;
        bool ok;
    };
    struct Regs {
        float4 v[NP];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        int r = r0 + (tid % TPR) * 4;
        s.ok = r < rows;
        if (!s.ok) r = 0;
        const int tap = (int)sbl_div((unsigned)r, g.fdC);
        s.c = r - tap * g.C;
        conv_tap_delta<false>(g, tap, s.dh, s.dw);
        s.kr = tid / TPR;
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
        const int hw = g.OH * g.OW;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int k = k0 + s.kr + ps * RPP;   // output pixel index (k >= kend: coordinates computed, never used)
            const int img = (int)sbl_div((unsigned)k, g.fdHW);
            const int rem = k - img * hw;
            const int oh = (int)sbl_div((unsigned)rem, g.fdOW), ow = rem - oh * g.OW;
            const int ih = oh * g.stride + s.dh, iw = ow * g.stride + s.dw;
            const bool in = (unsigned)ih < (unsigned)g.SH && (unsigned)iw < (unsigned)g.SW;
            // pixel index < 2^24 (host-checked): 24-bit multiplies run at full rate, 32-bit ones at a quarter
            const unsigned pix = __umul24(__umul24((unsigned)img, (unsigned)g.SH) + (unsigned)ih, (unsigned)g.SW) + (unsigned)iw;
            r.v[ps] = sbl_ld4(s.rs, s.ok && k < kend && in ? (__umul24(pix, (unsigned)g.C) + (unsigned)s.c) * 4u : SBL_OOB);
        }
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_mc<BR>(lds, r, tid); }
};

// ---- position-major 3x3 / stride-1 convolutions on small maps (ResNet layers 2-4: 11x11, 6x6 and 3x3 pixels).
// With pad 1 a border pixel sees only 4 or 6 of the 9 taps; on a 3x3 map 40 % (6x6: 21 %) of the im2col matrix is
// zero padding.  Ordering the GEMM rows position-major (row = pos * NIMG + img) makes the set of in-bounds taps
// (nearly) uniform per tile, so each workgroup contracts only over the taps its positions can reach; products with
// the padded zeros are skipped, every kept product is the same as before (bit-identical accumulation order per tap).
template <int BR, bool DGRAD>
struct ConvGatherPM {
    static constexpr bool kColSum = false;
    static constexpr bool kKC = true;
    static constexpr bool kDenseOperand = false;
    const float* p;
    ConvGeom g;
    int rows;   // NIMG*OH*OW
    unsigned long long taps;   // the workgroup's tap list, 4 bits each (set by sbl_conv_pm_kernel)
    struct State {
        sbl_rsrc rs;
        int base[BR / 64];
        unsigned mask[BR / 64];     // indexed by tap id kh*KW + kw
    };
    struct Regs {
        float4 v[BR / 64];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const int r = r0 + ps * 64 + (tid >> 2);
            const bool ok = r < rows;
            const unsigned rr = ok ? (unsigned)r : 0u;
            const int pos = (int)sbl_div(rr, g.fdNIMG);
            const int img = (int)rr - pos * g.NIMG;
            const int oh = (int)sbl_div((unsigned)pos, g.fdOW), ow = pos - oh * g.OW;
            s.base[ps] = (((img * g.SH + oh) * g.SW + ow) * g.C + (tid & 3) * 4) * 4;       // stride 1 both ways
            s.mask[ps] = ok ? conv_tap_mask<DGRAD, false>(g, oh, ow, g.KH * g.KW) : 0u;
        }
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
        const int t = (int)sbl_div((unsigned)k0, g.fdC);            // workgroup-uniform
        const int tap = (int)((taps >> (4 * (t & 15))) & 15ull);
        int dh, dw;
        conv_tap_delta<DGRAD, false>(g, tap, dh, dw);
        const int delta = ((dh * g.SW + dw) * g.C + (k0 - t * g.C)) * 4;
        const unsigned bit = k0 < kend ? (1u << tap) : 0u;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) r.v[ps] = sbl_ld4(s.rs, (s.mask[ps] & bit) ? (unsigned)(s.base[ps] + delta) : SBL_OOB);
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_kc<BR>(lds, r, tid); }
};

// Weight gradient, position-major: a tile of the (tap, ci) axis that lies inside ONE tap contracts only over the
// output pixels for which that tap is in bounds - a rectangle [oh_lo, oh_lo+nh) x [ow_lo, ow_lo+nw) of the map, all
// images: GEMM-k' = vp * NIMG + img with vp the index inside the rectangle.  PmRect is set per workgroup.
struct PmRect {
    int oh_lo, ow_lo, nw, dh, dw;    // source pixel = (oh + dh, ow + dw), always in bounds inside the rectangle
    float rnw;                       // 1 / nw
};
template <int BR>
struct DenseMCPM {            // dY^T: element (co, k') = dy[pixel(k')][co]
    static constexpr bool kColSum = false;
    static constexpr bool kKC = false;
    static constexpr bool kDenseOperand = false;
    const float* p;
    long ld;
    int rows;
    int NIMG, OH, OW;
    FastDiv fdNIMG;
    PmRect rc;
    static constexpr int TPR = BR / 4;
    static constexpr int RPP = 256 / TPR;
    static constexpr int NP = SBL_BK / RPP;
    struct State {
        sbl_rsrc rs;
        int c, kr;
    };
    struct Regs {
        float4 v[NP];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        s.c = r0 + (tid % TPR) * 4;
        s.kr = tid / TPR;
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int k = k0 + s.kr + ps * RPP;
            const int vp = (int)sbl_div((unsigned)k, fdNIMG), img = k - vp * NIMG;
            const int a = (int)(((float)vp + 0.5f) * rc.rnw), b = vp - a * rc.nw;      // vp / nw, exact for these small ints
            const unsigned pix = __umul24(__umul24((unsigned)img, (unsigned)OH) + (unsigned)(rc.oh_lo + a), (unsigned)OW) + (unsigned)(rc.ow_lo + b);
            r.v[ps] = sbl_ld4(s.rs, k < kend && s.c < rows ? (__umul24(pix, (unsigned)ld) + (unsigned)s.c) * 4u : SBL_OOB);
        }
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State& s) const { return s.c; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_mc<BR>(lds, r, tid); }
};
template <int BR>
struct ConvGatherMCPM {       // x gathered: element ((tap, ci), k') = x[img, oh + dh, ow + dw, ci]
    static constexpr bool kColSum = false;
    static constexpr bool kKC = false;
    static constexpr bool kDenseOperand = false;
    const float* p;
    ConvGeom g;
    int rows;   // KH*KW*C
    PmRect rc;
    static constexpr int TPR = BR / 4;
    static constexpr int RPP = 256 / TPR;
    static constexpr int NP = SBL_BK / RPP;
    struct State {
        sbl_rsrc rs;
        int c, kr;
        bool ok;
    };
    struct Regs {
        float4 v[NP];
    };
    __device__ __forceinline__ void init(State& s, int r0, int tid) const {
        s.rs = sbl_make_rsrc(p);
        int r = r0 + (tid % TPR) * 4;
        s.ok = r < rows;
        if (!s.ok) r = 0;
        s.c = r % g.C;
        s.kr = tid / TPR;
    }
    __device__ __forceinline__ void load(const State& s, int k0, int kend, Regs& r) const {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int k = k0 + s.kr + ps * RPP;
            const int vp = (int)sbl_div((unsigned)k, g.fdNIMG), img = k - vp * g.NIMG;
            const int a = (int)(((float)vp + 0.5f) * rc.rnw), b = vp - a * rc.nw;
            const unsigned pix = __umul24(__umul24((unsigned)img, (unsigned)g.SH) + (unsigned)(rc.oh_lo + a + rc.dh), (unsigned)g.SW) + (unsigned)(rc.ow_lo + b + rc.dw);
            r.v[ps] = sbl_ld4(s.rs, s.ok && k < kend ? (__umul24(pix, (unsigned)g.C) + (unsigned)s.c) * 4u : SBL_OOB);
        }
    }
    __device__ __forceinline__ void accum(const Regs&, float4&) const {}
    __device__ __forceinline__ int col(const State&) const { return 0; }
    __device__ __forceinline__ void store(float (*lds)[BR + 4], const State&, const Regs& r, int tid) const { sbl_store_mc<BR>(lds, r, tid); }
};

