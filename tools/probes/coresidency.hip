// Do two workgroups of W wavefronts, R registers per lane and L bytes of LDS share a CU?  Each workgroup spins for a fixed
// number of clock ticks; a grid of 2 x 256 workgroups takes as long as one of 256 if (and only if) the two co-reside.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/coresidency.hip -o tools/probes/coresidency.bin    run: tools/probes/coresidency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int THREADS, int REGS>
__global__ __launch_bounds__(THREADS) void spin(float* out, long ticks) {
    extern __shared__ float lds[];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[1] + s;
}
template <int THREADS, int REGS>
static void run(const char* name, size_t lds) {
    float* out;
    hipMalloc(&out, 4096 * sizeof(float));
    hipFuncSetAttribute((const void*)spin<THREADS, REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {256, 512, 768, 1024}) {
        hipLaunchKernelGGL((spin<THREADS, REGS>), dim3(grid), dim3(THREADS), lds, 0, out, 20000L);      // 200 us at 100 MHz
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((spin<THREADS, REGS>), dim3(grid), dim3(THREADS), lds, 0, out, 20000L);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        printf("%-40s lds %6zu KB  grid %4d: %.0f us\n", name, lds / 1024, grid, ms * 1e3);
    }
    hipFree(out);
}
int main() {
    run<256, 120>("4 waves, ~128 regs", 70 * 1024);
    run<384, 120>("6 waves, ~128 regs", 70 * 1024);
    run<384, 150>("6 waves, ~160 regs", 70 * 1024);
    run<384, 150>("6 waves, ~160 regs", 76 * 1024 + 512);
    run<512, 100>("8 waves, ~110 regs", 70 * 1024);
    run<768, 150>("12 waves, ~160 regs", 135 * 1024);
    return 0;
}
