// Exact split of fp32 values into bf16 "planes" and the plane-product lists of the split-bf16 MFMA modes (shared by the
// tile engine, bf16_tile.h, and the stem kernels, stem.hip).  See bf16_tile.h for the arithmetic.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int NT>
struct BfTerms;
template <>
struct BfTerms<1> {
    static constexpr int NPL = 1, N = 1;
    __device__ static constexpr int pa(int) { return 0; }
    __device__ static constexpr int pb(int) { return 0; }
};
template <>
struct BfTerms<3> {
    static constexpr int NPL = 2, N = 3;
    __device__ static constexpr int pa(int t) { return t == 1 ? 1 : 0; }     // (0,1) (1,0) (0,0): small terms first
    __device__ static constexpr int pb(int t) { return t == 0 ? 1 : 0; }
};
template <>
struct BfTerms<6> {
    static constexpr int NPL = 3, N = 6;
    __device__ static constexpr int pa(int t) { return t == 0 ? 0 : t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 0 : t == 4 ? 1 : 0; }
    __device__ static constexpr int pb(int t) { return t == 0 ? 2 : t == 1 ? 1 : t == 2 ? 0 : t == 3 ? 1 : t == 4 ? 0 : 0; }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair (round to nearest even, one v_cvt_pk_bf16_f32) and back to two fp32.  The low element comes
// back through v_perm_b32 rather than `pair << 16`: for the shift the compiler re-derives the low element with a second
// conversion instruction (only one half of the pair is "demanded").
__device__ __forceinline__ unsigned bf_pack2(f32x2 v) {
    bf16x2 p = {(__bf16)v.x, (__bf16)v.y};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ f32x2 bf_unpack2(unsigned u) {
    return f32x2{__builtin_bit_cast(float, __builtin_amdgcn_perm(u, 0u, 0x05040c0cu)), __builtin_bit_cast(float, u & 0xffff0000u)};
}

// four consecutive fp32 values -> NPL planes of four bf16 (8 bytes each); 9 VALU operations per pair of values for three
// planes (3 conversions, 2 x (shift, mask), 2 packed subtractions)
template <int NPL>
__device__ __forceinline__ void bf_split4(const float4& v, uint2 (&out)[NPL]) {
    const f32x2 x = {v.x, v.y}, y = {v.z, v.w};
    const unsigned a0 = bf_pack2(x), b0 = bf_pack2(y);
    out[0] = make_uint2(a0, b0);
    if constexpr (NPL > 1) {
        const f32x2 rx = x - bf_unpack2(a0), ry = y - bf_unpack2(b0);
        const unsigned a1 = bf_pack2(rx), b1 = bf_pack2(ry);
        out[1] = make_uint2(a1, b1);
        if constexpr (NPL > 2) out[2] = make_uint2(bf_pack2(rx - bf_unpack2(a1)), bf_pack2(ry - bf_unpack2(b1)));
    }
}

