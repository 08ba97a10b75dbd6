// Shared device/host helpers for libsbl_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sbl_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SBL_WAVE 64

// ---- error plumbing: 0 on success, hipError_t / negative code otherwise; text in sbl_last_error()
void sbl_set_error(const char* fmt, ...);
#define SBL_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) {                                          \
            sbl_set_error(__VA_ARGS__);                         \
            return SBL_ERR_INVALID;                             \
        }                                                       \
    } while (0)
#define SBL_LAUNCH_CHECK(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            sbl_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                      \
        }                                                                         \
    } while (0)
#define SBL_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            sbl_set_error("%s failed: %s", #call, hipGetErrorString(e__));              \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

// ---- optional in-kernel wall-clock stamps (bench instrumentation; off unless sbl_profile_begin() was called)
// slot[0] = min over workgroups of the start time, slot[1] = max of the end time, 100 MHz s_memrealtime ticks.
// Works inside hipGraph replays (the slot pointer is baked into the captured launch).
unsigned long long* sbl_next_stamp_slot(int kernel_id);   // host side; nullptr when profiling is off
__device__ __forceinline__ void sbl_stamp_begin(unsigned long long* slot) {
    if (slot && threadIdx.x == 0) atomicMin(slot, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void sbl_stamp_end(unsigned long long* slot) {
    if (slot && threadIdx.x == 0) atomicMax(slot + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
#define SBL_KID_SKINNY 1
#define SBL_KID_TILED64 2
#define SBL_KID_TILED128 3
#define SBL_KID_CONV_FWD 4
#define SBL_KID_CONV_DGRAD 5
#define SBL_KID_CONV_WGRAD 6
#define SBL_KID_SEG_WGRAD 7
#define SBL_KID_STEM 8
#define SBL_KID_ATTENTION 9

// ---- ragged row batches: a "run" of decoder steps whose inputs are all known is processed as one batch of
// segments; segment s holds B sequences of length L[s], rows [row_off[s], row_off[s] + B*L[s]) in (b, l) order.
#define SBL_MAX_SEG 16
struct SegDesc {
    int nseg;
    int L[SBL_MAX_SEG];
    int row_off[SBL_MAX_SEG];
    int p_off[SBL_MAX_SEG];     // offset (floats) of the segment's attention-probability block
    int qtile;                  // 0, or L: the "segments" are 16-row QUERY TILES of one self-attention over L rows per batch entry
                                // (row_off / p_off are then offsets inside a sequence / inside a (head, batch) probability block)
};
// host: build from an array of lengths; returns total rows (or -1 on bad input)
static inline long sbl_make_segs(SegDesc& d, const int* seg_L, int nseg, int B, int H, int Lk_fixed) {
    if (nseg < 1 || nseg > SBL_MAX_SEG || !seg_L) return -1;
    d.nseg = nseg;
    d.qtile = 0;
    long rows = 0, p = 0;
    for (int s = 0; s < SBL_MAX_SEG; ++s) {
        d.L[s] = s < nseg ? seg_L[s] : 0;
        d.row_off[s] = (int)rows;
        d.p_off[s] = (int)p;
        if (s < nseg) {
            if (seg_L[s] < 1) return -1;
            rows += (long)B * seg_L[s];
            p += (long)H * B * seg_L[s] * (Lk_fixed > 0 ? Lk_fixed : seg_L[s]);
            if (rows > (1L << 30) || p > (1L << 30)) return -1;
        }
    }
    return rows;
}
// device: segment containing row r
__device__ __forceinline__ int sbl_seg_of_row(const SegDesc& d, int r, int B) {
    int s = 0;
#pragma unroll
    for (int t = 1; t < SBL_MAX_SEG; ++t)
        if (t < d.nseg && r >= d.row_off[t]) s = t;
    return s;
}

static inline int sbl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool sbl_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- wave-level reductions (64-wide wavefront; no LDS)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- counter-based RNG for dropout: one 32-bit draw per (seed, stream offset, element index).
// The seed lives in device memory so a captured hipGraph draws fresh masks on every replay
// (a one-thread kernel bumps it per step); fwd and bwd regenerate the same mask from it.
__device__ __forceinline__ uint32_t sbl_rand_u32(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (offset + 1) + idx * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
// keep-decision with drop probability p: threshold = p * 2^32
__device__ __forceinline__ bool sbl_keep(uint64_t seed, uint64_t offset, uint64_t idx, uint32_t thresh) {
    return sbl_rand_u32(seed, offset, idx) >= thresh;
}
static inline uint32_t sbl_drop_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}
