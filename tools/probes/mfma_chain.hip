// Issue/latency probe for v_mfma_f32_32x32x16_bf16 on gfx950: cycles per MFMA for NACC independent accumulator chains per
// wave and W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void chain(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd) {
    float* out; unsigned long long* cyc;
    const int threads = 256 * waves_per_simd;      // 4 SIMDs x waves
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    chain<NACC><<<256, threads>>>(out, cyc, 10);
    hipEventRecord(a);
    chain<NACC><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;      // MFMAs per wave
    printf("acc chains %d, waves/SIMD %d: %.1f counter ticks and %.2f ns per MFMA per wave; per SIMD %.2f ns per MFMA (%.0f TFLOP/s on 256 CUs)\n",
           NACC, waves_per_simd, c / n, ms * 1e6 / n, ms * 1e6 / n / waves_per_simd, 256.0 * 4 * 32768 / (ms * 1e6 / n / waves_per_simd) / 1e3);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1);
    run<1>(2); run<1>(3); run<2>(2); run<1>(4);
    return 0;
}
