// What each stage of the 64x64 split-bf16 tile loop costs: the same kernel built with one stage removed at a time
// (SBL_ABL bits, csrc/bf16_tile.h).  Build one binary per mask:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSBL_ABL=<mask> tools/probes/tile_ablate.hip -o tools/probes/tile_ablate_<mask>
// Run: tile_ablate_<mask> M N K  (A is M x K, B is N x K, both k-contiguous; 6-product mode)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../sbl_for_multilingual_lip_reading_amd/csrc/mfma_gemm.h"
int g_sbl_prec = 6;
// PROBE_LINES: every K slab of a row starts its own 128-byte line (stored row = 2K floats, slab s at float 32 s): no two
// load instructions ever touch the same line
template <int BR>
struct DenseKCLines : DenseKC<BR, true> {
    using Base = DenseKC<BR, true>;
    __device__ __forceinline__ void load(const typename Base::State& s, int k0, int kend, typename Base::Regs& r) const {
        const int k = k0 + s.kq;
#pragma unroll
        for (int ps = 0; ps < BR / 64; ++ps) r.v[ps] = sbl_ld4(s.rs, s.ro[ps] != SBL_OOB && k < kend ? s.ro[ps] + (unsigned)k0 * 8 : SBL_OOB);
    }
};
#ifndef PBM
#define PBM 64
#define PBN 64
#endif
__global__ void fill(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.f);
    }
}
#ifdef PROBE_LINES
using LDA = DenseKCLines<PBM>; using LDB = DenseKCLines<PBN>;
#define LDMUL 2
#else
using LDA = DenseKC<PBM, true>; using LDB = DenseKC<PBN, true>;
#define LDMUL 1
#endif
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4 * LDMUL); hipMalloc(&B, (size_t)N * K * 4 * LDMUL); hipMalloc(&C, (size_t)M * N * 4);
    fill<<<1024, 256>>>(A, (size_t)M * K * LDMUL); fill<<<1024, 256>>>(B, (size_t)N * K * LDMUL);
    LDA al{}; LDB bl{};
    al.p = A; al.ld = (long)K * LDMUL; al.rows = M; bl.p = B; bl.ld = (long)K * LDMUL; bl.rows = N;
    EpiStore<0, false, false> epi{};
    epi.C = C; epi.ldc = N;
    SplitCtl sc{};
    dim3 grid(M / PBM, N / PBN, 1);
    auto launch = [&] { sbl_mfma_gemm_kernel<LDA, LDB, EpiStore<0, false, false>, PBM, PBN, 1, 2, 6><<<grid, 256>>>(al, bl, epi, sc, M, N, K, K, -1); };
    for (int i = 0; i < 5; ++i) launch();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("ABL=%2d lines=%d tile %dx%d  %dx%dx%d  %.1f us\n", SBL_ABL, LDMUL - 1, PBM, PBN, M, N, K, ms * 1e3 / 50);
    return 0;
}
