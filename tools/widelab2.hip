// widelab2.hip -- what each part of gemm_wide_kernel<f16, true, true, true> (weights from the decode copy, raw x by LDS DMA) costs: the product kernel against
// builds with one thing removed (-DGPTQ_WIDE_ABL=<bits>: 1 no dequant math, 2 no A-fragment LDS reads after a step's first, 4 no barrier, 16 no x DMA,
// 32 no weight loads; wrong results by construction), M = 4096 on 4096 x 4096 (or argv: M K N).  Measurement tool, not product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 [-DGPTQ_WIDE_ABL=n] -I autogptq_amd/csrc -I include tools/widelab2.hip autogptq_amd/csrc/utils.o -o tools/widelab2_abl<n>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "gemm_wide.hip"
using namespace gptq;
namespace gptq { hipError_t launch_permute_rows16(const void*, const int32_t*, int, int, void*, hipStream_t, bool) { return hipErrorNotSupported; } }   // utils.o refers to it (gemm.hip has it)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned v = (unsigned)i * 2654435761u + seed; v ^= v >> 15; v *= 2246822519u; v ^= v >> 13; p[i] = v; }
}
__global__ void fill_f16(_Float16* p, size_t n, float lo, float hi) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned v = (unsigned)i * 2654435761u + 7u; v ^= v >> 15; v *= 2246822519u; v ^= v >> 13; p[i] = (_Float16)(lo + (hi - lo) * (v & 0xffff) / 65535.f); }
}
int main(int argc, char** argv) {
    const int M = argc >= 4 ? atoi(argv[1]) : 4096, K = argc >= 4 ? atoi(argv[2]) : 4096, N = argc >= 4 ? atoi(argv[3]) : 4096, nl = 6;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t qw_b = (size_t)K / 8 * N * 4, qz_b = (size_t)(K / 128) * N / 8 * 4, sc_b = (size_t)(K / 128) * N * 2, cb = (size_t)(K / 128) * 48 * (N / 16);
    unsigned *qw, *qz, *tq; _Float16 *sc, *x, *out; char* cst;
    CK(hipMalloc(&qw, qw_b * nl)); CK(hipMalloc(&tq, qw_b * nl)); CK(hipMalloc(&qz, qz_b * nl)); CK(hipMalloc(&sc, sc_b * nl)); CK(hipMalloc(&cst, cb * nl));
    CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2 * nl));
    fill<<<2048, 256, 0, st>>>(qw, qw_b * nl / 4, 1u); fill<<<256, 256, 0, st>>>(qz, qz_b * nl / 4, 2u);
    fill_f16<<<256, 256, 0, st>>>(sc, sc_b * nl / 2, 0.002f, 0.0022f); fill_f16<<<2048, 256, 0, st>>>(x, (size_t)M * K, -0.5f, 0.5f);
    gptq_layer_t Ls[8];
    for (int i = 0; i < nl; ++i) {
        gptq_layer_t L{};
        L.K = K; L.N = N; L.bits = 4; L.group_size = 128; L.dtype = GPTQ_F16; L.zero_mode = GPTQ_ZERO_WRAP;
        L.qweight = qw + (size_t)i * qw_b / 4; L.qzeros = qz + (size_t)i * qz_b / 4; L.scales = sc + (size_t)i * sc_b / 2;
        CK(launch_prepack_decode(L.qweight, L.qzeros, L.scales, K, N, 4, 128, L.zero_mode, tq + (size_t)i * qw_b / 4, cst + (size_t)i * cb, st));
        L.qweight_tiled = tq + (size_t)i * qw_b / 4; L.qconst_tiled = cst + (size_t)i * cb; L.tiled_cols = 16;
        Ls[i] = L;
    }
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto all = [&]() { for (int i = 0; i < nl; ++i) CK(launch_gemm_wide(Ls[i], Ls[i].qweight, x, out + (size_t)i * M * N, M, true, st, true)); };
    for (int w = 0; w < 5; ++w) all();
    CK(hipStreamSynchronize(st));
    double best = 1e30, sum = 0; const int rounds = 8, reps = 3;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k) all();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * nl);
        best = std::min(best, us); sum += us;
    }
#ifndef GPTQ_WIDE_ABL
#define GPTQ_WIDE_ABL 0
#endif
    printf("abl=%2d  M=%d K=%d N=%d: min %7.2f us %7.1f TFLOP/s   mean %7.2f us %7.1f TFLOP/s\n", GPTQ_WIDE_ABL, M, K, N, best, 2.0 * M * K * N / best / 1e6, sum / rounds, 2.0 * M * K * N / (sum / rounds) / 1e6);
    return 0;
}
