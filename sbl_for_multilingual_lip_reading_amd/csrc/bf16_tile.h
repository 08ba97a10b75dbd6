// bf16 MFMA body of the tile engine (included by mfma_gemm.h; same loaders, split-K and epilogues as the fp32 body).
//
// Every fp32 operand element x is split on the way into LDS into NPL bf16 "planes":
//     p0 = bf16(x)   p1 = bf16(x - p0)   p2 = bf16(x - p0 - p1)        (round-to-nearest-even, v_cvt_pk_bf16_f32)
// Three planes hold 27 significand bits >= fp32's 24, so x = p0 + p1 + p2 exactly (non-finite x gives NaN planes).
// A product a*b is then the sum of plane products, each of which v_mfma_f32_32x32x16_bf16 forms exactly (8 x 8
// significand bits) and accumulates in fp32:
//     NT = 6 : a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0          dropped terms <= 2^-26 |ab|: fp32-grade results
//     NT = 3 : a0b0 + a0b1 + a1b0                               ~2^-17 |ab| per product
//     NT = 1 : a0b0                                             plain bf16 inputs (BASELINE config 5 "mixed bf16")
// at 6 / 3 / 1 MFMAs of 32 cycles per 32x32x16 block, against 8 fp32 MFMAs of 64 cycles (mfma_f32_32x32x2_f32).
//
// LDS images, one per plane and 16-deep K slab:
//   k-contiguous loaders (kKC): [row][16 k] bf16 = 32 B per row, the two 16-byte halves of a row swapped on odd
//     8-row groups, so both the 8-byte stores (16 consecutive lanes cover 128 contiguous bytes) and the ds_read_b128
//     operand reads (lane l: row l&31, k = 8*(l>>5)..+7) are bank-conflict free.
//   m-contiguous loaders: [16 k][BR m] bf16 with 64 B of padding per row; the MFMA operand (8 consecutive k of one row)
//     comes out of two ds_read_b64_tr_b16 transposed reads (4 k x 16 m blocks; conflict free with that padding).
#pragma once

#include "bf16_split.h"

template <int BR, bool KC>
struct BfImage {
    static constexpr int ROW = KC ? 32 : (BR * 2 + 64);       // bytes per image row
    static constexpr int BYTES = KC ? BR * 32 : 16 * ROW;     // one plane of one 16-deep slab
};

// registers of one loader slab -> its LDS planes (img = plane 0 of the slab; planes are BYTES apart)
template <class L, int BR, int NPL>
__device__ __forceinline__ void bf_store(unsigned char* img, const typename L::Regs& r, int tid) {
    using I = BfImage<BR, L::kKC>;
    if constexpr (L::kKC) {
        // thread: rows ps*64 + (tid>>2), k = (tid&3)*4 .. +3
        const int kq = (tid & 3) * 4;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) {
            const int row = ps * 64 + (tid >> 2);
            const int off = row * 32 + ((((kq >> 3) ^ (tid >> 5)) & 1) << 4) + ((kq & 4) << 1);
            uint2 p[NPL];
            bf_split4<NPL>(r.v[ps], p);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint2*>(img + pl * I::BYTES + off) = p[pl];
        }
    } else {
        // thread: m = (tid % TPR)*4 .. +3, k rows tid/TPR + ps*RPP
        constexpr int TPR = BR / 4, RPP = 256 / TPR, NP = SBL_BK / RPP;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int off = (tid / TPR + ps * RPP) * I::ROW + (tid % TPR) * 8;
            uint2 p[NPL];
            bf_split4<NPL>(r.v[ps], p);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint2*>(img + pl * I::BYTES + off) = p[pl];
        }
    }
}

// MFMA operand of the 32-row block at row base rb: lane l gets row rb + (l&31), k = 8*(l>>5) .. +7
template <int BR, bool KC>
__device__ __forceinline__ bf16x8 bf_frag(const unsigned char* img, int rb, int lane) {
    using I = BfImage<BR, KC>;
    if constexpr (KC) {
        const int row = rb + (lane & 31);
        return *reinterpret_cast<const bf16x8*>(img + row * 32 + ((((lane >> 5) ^ (row >> 3)) & 1) << 4));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const unsigned char* a = img + (8 * (g >> 1) + q) * I::ROW + (rb + 16 * (g & 1) + 4 * p) * 2;
        typedef __attribute__((address_space(3))) bf16x4* lds_p;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a + 4 * I::ROW));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// SBL_ABL (tools/probes/tile_ablate.hip only; never set in the library build): bit 0 drops the MFMAs, bit 1 the LDS fragment
// reads, bit 2 the split + LDS stores, bit 3 the global loads of the K loop - what each stage costs is the time that goes
// away with it.
#ifndef SBL_ABL
#define SBL_ABL 0
#endif
// NH = 2 (64x64 tiles, no split-K over workgroups): the workgroup has 512 threads = two groups of four wavefronts; group h
// stages and multiplies the slabs 2j + h of the tile's K range in its own pair of LDS buffers, with its own accumulators, and
// the two partial tiles meet once in LDS before the epilogue.  Same global traffic as one group, half the barrier intervals
// per tile, twice the loads and MFMAs in flight per tile: these launches have only ~2 tiles per CU and every tile is a serial
// chain of 32-128 slab iterations, so the chain length, not the chip, bounds them.
template <class AL, class BL, class EPI, int BM, int BN, int KU, int WN, int NT, int NH = 1>
__device__ __forceinline__ void sbl_gemm_tile_bf(const AL& al, const BL& bl, const EPI& epi, const SplitCtl& sc, int M, int N,
                                                 int m0, int n0, int kbeg, int kend, int tile, int z, int nz,
                                                 bool colsum_tile) {
    using T = BfTerms<NT>;
    constexpr int NPL = T::NPL;
    using IA = BfImage<BM, AL::kKC>;
    using IB = BfImage<BN, BL::kKC>;
    constexpr int A_SLAB = NPL * IA::BYTES, B_SLAB = NPL * IB::BYTES;
    constexpr int A_BUF = KU * A_SLAB, BUF = KU * (A_SLAB + B_SLAB);
    constexpr int MK = KU * SBL_BK;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int RD = KU != 1 ? 1 : (TM * TN == 1 ? 4 : 2);      // register ring depth (below)
    static_assert(NH == 1 || (NH == 2 && RD == 4), "wave-group K split: 64x64 tiles only");
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[NH * 2 * BUF];
    const int half = NH == 2 ? (int)(threadIdx.x >> 8) : 0;
    unsigned char* smem = smem_all + half * (2 * BUF);
    constexpr int KS = NH * SBL_BK;              // K stride between consecutive slabs of one wave group
    const int hoff = half * SBL_BK;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    sbl_stamp_begin(sc.stamp);

    typename AL::State sa;
    typename BL::State sb;
    typename AL::Regs ra[KU];
    typename BL::Regs rb[KU];
    al.init(sa, m0, tid);
    bl.init(sb, n0, tid);
    const bool do_colsum = AL::kColSum && sc.a_colsum != nullptr && colsum_tile;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * (BM / WM), brow = wn * (BN / WN);
    auto compute = [&](const unsigned char* base) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            bf16x8 a[NPL][TM], b[NPL][TN];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (SBL_ABL & 2) { for (int e = 0; e < 8; ++e) a[pl][i][e] = (__bf16)(float)(lane + e); }
                    else a[pl][i] = bf_frag<BM, AL::kKC>(base + u * A_SLAB + pl * IA::BYTES, arow + i * 32, lane);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (SBL_ABL & 2) { for (int e = 0; e < 8; ++e) b[pl][j][e] = (__bf16)(float)(lane - e); }
                    else b[pl][j] = bf_frag<BN, BL::kKC>(base + A_BUF + u * B_SLAB + pl * IB::BYTES, brow + j * 32, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t = 0; t < T::N; ++t) {
                        if (SBL_ABL & 1) asm volatile("" ::"v"(a[T::pa(t)][i]), "v"(b[T::pb(t)][j]));
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[T::pa(t)][i], b[T::pb(t)][j], acc[i][j], 0, 0, 0);
                    }
        }
    };
    // Register rings keep the global loads of several slabs in flight; loads past the end of K are issued anyway (out-of-range
    // offset: zeros, no memory access), so the loop bodies have no branch around a load and the compiler keeps counted
    // vmcnt waits.
    //  * 64x64 tiles (one accumulator per wave, 8 staging registers per slab; one or two workgroups per CU on the
    //    4352-row products, 6 MFMAs per slab): four sets - nothing else hides a load round trip.  (Two slabs per LDS buffer
    //    and per barrier were measured too: 48 KB of LDS per workgroup costs more residency than the barriers cost.)
    //  * larger tiles: two sets, one slab per barrier, and the split of the next slab (VALU) spread through the shadows of
    //    this slab's MFMAs with scheduling-group barriers.
    if constexpr (RD == 4) {
        typename AL::Regs qa[RD];
        typename BL::Regs qb[RD];
#pragma unroll
        for (int q = 0; q < RD; ++q) {
            al.load(sa, kbeg + hoff + q * KS, kend, qa[q]);
            bl.load(sb, kbeg + hoff + q * KS, kend, qb[q]);
        }
        if (do_colsum) al.accum(qa[0], cs);
        bf_store<AL, BM, NPL>(smem, qa[0], tid);
        bf_store<BL, BN, NPL>(smem + A_BUF, qb[0], tid);
        al.load(sa, kbeg + hoff + RD * KS, kend, qa[0]);
        bl.load(sb, kbeg + hoff + RD * KS, kend, qb[0]);
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += RD * KS) {
#pragma unroll
            for (int q = 0; q < RD; ++q) {
                const int k = k0 + q * KS;              // (workgroup-uniform: the first slab of this barrier interval)
                if (k >= kend) break;                   // slab k + hoff in LDS buffer q & 1; ring slot (q + 1) % RD holds the next
                unsigned char* nb = smem + ((q + 1) & 1) * BUF;
                if (do_colsum) al.accum(qa[(q + 1) % RD], cs);
                if (!(SBL_ABL & 4)) {
                    bf_store<AL, BM, NPL>(nb, qa[(q + 1) % RD], tid);
                    bf_store<BL, BN, NPL>(nb + A_BUF, qb[(q + 1) % RD], tid);
                } else {
                    const float4 ka = qa[(q + 1) % RD].v[0], kb = qb[(q + 1) % RD].v[0];
                    asm volatile("" ::"v"(ka.x), "v"(ka.y), "v"(ka.z), "v"(ka.w), "v"(kb.x), "v"(kb.y), "v"(kb.z), "v"(kb.w));
                }
                if (!(SBL_ABL & 8)) {
                    al.load(sa, k + hoff + (RD + 1) * KS, kend, qa[(q + 1) % RD]);
                    bl.load(sb, k + hoff + (RD + 1) * KS, kend, qb[(q + 1) % RD]);
                }
                compute(smem + (q & 1) * BUF);
                __syncthreads();
            }
        }
        if constexpr (NH == 2) {
            // (every wavefront is past its last LDS operand read: the loop ends on a barrier)
            float* red = reinterpret_cast<float*>(smem_all);
            if (half == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[r * 256 + tid] = acc[0][0][r];
            }
            __syncthreads();
            if (half == 1) {
                if (do_colsum) {      // this group's share of the bias-gradient column sums
                    const int c = al.col(sa);
                    if (c + 0 < M) atomicAdd(sc.a_colsum + c + 0, cs.x);
                    if (c + 1 < M) atomicAdd(sc.a_colsum + c + 1, cs.y);
                    if (c + 2 < M) atomicAdd(sc.a_colsum + c + 2, cs.z);
                    if (c + 3 < M) atomicAdd(sc.a_colsum + c + 3, cs.w);
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += red[r * 256 + tid];
        }
    } else if constexpr (RD == 2) {
        typename AL::Regs qa[RD];
        typename BL::Regs qb[RD];
#pragma unroll
        for (int q = 0; q < RD; ++q) {
            al.load(sa, kbeg + q * SBL_BK, kend, qa[q]);
            bl.load(sb, kbeg + q * SBL_BK, kend, qb[q]);
        }
        if (do_colsum) al.accum(qa[0], cs);
        bf_store<AL, BM, NPL>(smem, qa[0], tid);
        bf_store<BL, BN, NPL>(smem + A_BUF, qb[0], tid);
        al.load(sa, kbeg + RD * SBL_BK, kend, qa[0]);
        bl.load(sb, kbeg + RD * SBL_BK, kend, qb[0]);
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += RD * SBL_BK) {
#pragma unroll
            for (int q = 0; q < RD; ++q) {          // an odd slab count runs one slab of zeros: no branch inside the body
                const int k = k0 + q * SBL_BK;      // slab in LDS buffer q; ring slot q ^ 1 holds slab k + 16
                unsigned char* nb = smem + (q ^ 1) * BUF;
                if (do_colsum) al.accum(qa[q ^ 1], cs);
                bf_store<AL, BM, NPL>(nb, qa[q ^ 1], tid);
                bf_store<BL, BN, NPL>(nb + A_BUF, qb[q ^ 1], tid);
                al.load(sa, k + (RD + 1) * SBL_BK, kend, qa[q ^ 1]);
                bl.load(sb, k + (RD + 1) * SBL_BK, kend, qb[q ^ 1]);
                compute(smem + q * BUF);
                constexpr int NM = TM * TN * T::N;
                constexpr int NV = (2 * NPL * ((BM + BN) * SBL_BK / 256) + 12 + NM - 1) / NM;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                }
                __syncthreads();
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            al.load(sa, kbeg + u * SBL_BK, kend, ra[u]);
            bl.load(sb, kbeg + u * SBL_BK, kend, rb[u]);
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (do_colsum) al.accum(ra[u], cs);
            bf_store<AL, BM, NPL>(smem + u * A_SLAB, ra[u], tid);
            bf_store<BL, BN, NPL>(smem + A_BUF + u * B_SLAB, rb[u], tid);
        }
        __syncthreads();
        int cur = 0;
        for (int k0 = kbeg; k0 < kend; k0 += MK) {
            const bool has_next = (k0 + MK) < kend;
            if (has_next) {
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    al.load(sa, k0 + MK + u * SBL_BK, kend, ra[u]);
                    bl.load(sb, k0 + MK + u * SBL_BK, kend, rb[u]);
                }
            }
            compute(smem + cur * BUF);
            if (has_next) {
                unsigned char* nb = smem + (cur ^ 1) * BUF;
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    if (do_colsum) al.accum(ra[u], cs);
                    bf_store<AL, BM, NPL>(nb + u * A_SLAB, ra[u], tid);
                    bf_store<BL, BN, NPL>(nb + A_BUF + u * B_SLAB, rb[u], tid);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    sbl_tile_finish<AL, EPI, BM, BN, WN, TM, TN>(al, sa, epi, sc, acc, cs, do_colsum, M, N, m0, n0, tile, z, nz);
}
