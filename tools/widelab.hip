// widelab.hip -- the 128 x 512 prefill kernel (csrc/gemm_wide.hip) against the 128 x 256 product kernel (csrc/gemm.hip): bit comparison of the outputs
// (same MFMA k order, one K group: the results must be identical) and interleaved timing.  Measurement tool, not product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I autogptq_amd/csrc -I include -c tools/widelab.hip -o /tmp/widelab.o && hipcc --offload-arch=gfx950 /tmp/widelab.o autogptq_amd/csrc/utils.o autogptq_amd/csrc/gemm_mid.o -o tools/widelab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gemm.hip"
#include "gemm_wide.hip"
using namespace gptq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7) ^ seed;
        v ^= v << 13; v ^= v >> 17; v ^= v << 5;
        p[i] = v;
    }
}
__global__ void fill_f16(f16* p, size_t n, float lo, float hi) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)(i * 2246822519u) ^ 0x9e3779b9u; v ^= v >> 15; v *= 2654435761u; v ^= v >> 13;
        p[i] = (f16)(lo + (hi - lo) * (float)(v & 0xffff) / 65536.f);
    }
}
__global__ void iota_perm(int* p, int K) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) p[i] = (int)(((long long)i * 2731 + 17) % K);
}
__global__ void diff_count(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* out) {
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += a[i] != b[i];
    if (bad) atomicAdd(out, bad);
}

int main(int argc, char** argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(init_gemm_device());
    struct Shape { int M, K, N; };
    std::vector<Shape> shapes = {{4096, 4096, 4096}, {4096, 4096, 11008}, {4096, 11008, 4096}, {2048, 4096, 11008}, {2048, 4096, 4096}, {8192, 4096, 4096}, {4000, 4096, 4128}};
    if (argc >= 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
    for (auto s : shapes) {
        const int M = s.M, K = s.K, N = s.N;
        const size_t qw_b = (size_t)K / 8 * N * 4, qz_b = (size_t)(K / 128) * N / 8 * 4, sc_b = (size_t)(K / 128) * N * 2;
        const int nl = 4;
        unsigned *qw, *qz; f16 *sc, *x, *out, *out2, *bias; char* ws; int* perm; unsigned long long* bad;
        CK(hipMalloc(&qw, qw_b * nl)); CK(hipMalloc(&qz, qz_b * nl)); CK(hipMalloc(&sc, sc_b * nl)); CK(hipMalloc(&bias, (size_t)N * 2));
        CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2)); CK(hipMalloc(&out2, (size_t)M * N * 2)); CK(hipMalloc(&perm, (size_t)K * 4));
        CK(hipMalloc(&bad, 8));
        const size_t ws_b = WS_HEADER_BYTES + (size_t)M * K * 2 + (size_t)2 * M * N * 4 + 4096;
        CK(hipMalloc(&ws, ws_b));
        CK(hipMemset(ws, 0, WS_HEADER_BYTES));
        fill<<<2048, 256, 0, st>>>(qw, qw_b * nl / 4, 1u);
        fill<<<256, 256, 0, st>>>(qz, qz_b * nl / 4, 2u);
        fill_f16<<<256, 256, 0, st>>>(sc, sc_b * nl / 2, 0.002f, 0.0022f);
        fill_f16<<<2048, 256, 0, st>>>(x, (size_t)M * K, -0.5f, 0.5f);
        fill_f16<<<64, 256, 0, st>>>(bias, (size_t)N, -0.1f, 0.1f);
        iota_perm<<<64, 256, 0, st>>>(perm, K);
        CK(hipStreamSynchronize(st));
        printf("== M=%d K=%d N=%d : %.2f GFLOP per launch, %d wide tiles, %d narrow tiles\n", M, K, N, 2.0 * M * K * N / 1e9, (M + 127) / 128 * ((N + 511) / 512), (M + 127) / 128 * ((N + 255) / 256));
        gptq_layer_t L{};
        L.K = K; L.N = N; L.bits = 4; L.group_size = 128; L.dtype = GPTQ_F16; L.zero_mode = GPTQ_ZERO_WRAP; L.bias = bias;
        gptq_tuning_t t1{}; t1.path = 3; t1.reserved[3] = 6;       // one K group: the same accumulation order as the wide kernel
        gptq_tuning_t td{}; td.path = 3;                           // the planner's default (two K groups where it uses them)
        auto layer = [&](int i, bool act) { gptq_layer_t Li = L; Li.qweight = qw + (size_t)i * qw_b / 4; Li.qzeros = qz + (size_t)i * qz_b / 4; Li.scales = sc + (size_t)i * sc_b / 2;
                                            if (act) { Li.g_idx = perm; Li.perm = perm; Li.qweight_seq = Li.qweight; } return Li; };
        // ---- bit comparison: plain layer
        {
            gptq_layer_t Li = layer(1, false);
            GemmPlan pl = plan_gemm(Li, M, &t1);
            CK(launch_gemm(Li, pl, x, out, M, ws, ws + WS_HEADER_BYTES, st));
            if (!wide_gemm_ok(Li, M, false, false)) { printf("  wide kernel not applicable\n"); continue; }
            CK(launch_gemm_wide(Li, Li.qweight, x, out2, M, false, st));
            CK(hipMemsetAsync(bad, 0, 8, st));
            diff_count<<<1024, 256, 0, st>>>((const unsigned short*)out, (const unsigned short*)out2, (size_t)M * N, bad);
            unsigned long long hb; CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            printf("  plain: %llu of %zu outputs differ from gemm_kernel<4, f16, 4, 64> (one K group)\n", hb, (size_t)M * N);
        }
        // ---- act-order: permuted x in slot order (the pre-pass), DMA staging
        {
            gptq_layer_t Li = layer(2, true);
            GemmPlan pl = plan_gemm(Li, M, &t1);
            CK(launch_gemm(Li, pl, x, out, M, ws, ws + WS_HEADER_BYTES, st));
            // the pre-pass left the permuted, slot-ordered x at the front of the workspace body
            gptq_layer_t Lw = Li; 
            CK(launch_permute_rows16(x, perm, M, K, ws + WS_HEADER_BYTES, st, true));
            CK(launch_gemm_wide(Lw, Li.qweight_seq, ws + WS_HEADER_BYTES, out2, M, true, st));
            CK(hipMemsetAsync(bad, 0, 8, st));
            diff_count<<<1024, 256, 0, st>>>((const unsigned short*)out, (const unsigned short*)out2, (size_t)M * N, bad);
            unsigned long long hb; CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            printf("  act-order (DMA-staged x): %llu of %zu outputs differ (plan xslot=%d glds=%d kg=%d)\n", hb, (size_t)M * N, (int)pl.xslot, (int)pl.glds, pl.kg);
        }
        // ---- timing, interleaved rounds
        struct V { const char* name; int kind; double us; };
        std::vector<V> vs = {{"gemm_kernel 128x256 (planner default)", 0, 1e30}, {"gemm_kernel 128x256 (one K group)", 1, 1e30}, {"gemm_wide 128x512", 2, 1e30},
                             {"act-order: permute + gemm_kernel (default)", 3, 1e30}, {"act-order: permute + gemm_wide (DMA)", 4, 1e30}};
        auto launch_all = [&](V& v) {
            for (int i = 0; i < nl; ++i) {
                gptq_layer_t Li = layer(i, v.kind >= 3);
                if (v.kind == 0 || v.kind == 1 || v.kind == 3) { GemmPlan pl = plan_gemm(Li, M, v.kind == 1 ? &t1 : &td); CK(launch_gemm(Li, pl, x, out, M, ws, ws + WS_HEADER_BYTES, st)); }
                else if (v.kind == 2) CK(launch_gemm_wide(Li, Li.qweight, x, out2, M, false, st));
                else { CK(launch_permute_rows16(x, perm, M, K, ws + WS_HEADER_BYTES, st, true)); CK(launch_gemm_wide(Li, Li.qweight_seq, ws + WS_HEADER_BYTES, out2, M, true, st)); }
            }
        };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 10; ++w) for (auto& v : vs) launch_all(v);
        CK(hipStreamSynchronize(st));
        const int reps = 3;
        for (int round = 0; round < 5; ++round)
            for (auto& v : vs) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) launch_all(v);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                v.us = std::min(v.us, ms * 1e3 / (reps * nl));
            }
        for (auto& v : vs) printf("  %9.2f us  %8.1f TFLOP/s  %s\n", v.us, 2.0 * M * K * N / v.us / 1e6, v.name);
        fflush(stdout);
        CK(hipFree(qw)); CK(hipFree(qz)); CK(hipFree(sc)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(out2)); CK(hipFree(ws)); CK(hipFree(perm)); CK(hipFree(bias)); CK(hipFree(bad));
    }
    return 0;
}
