// gemmlab.hip -- timing harness for the PRODUCT MFMA prefill kernel (measurement tool, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -DGPTQ_GEMM_ABLATIONS -I autogptq_amd/csrc -I include -c tools/gemmlab.hip -o /tmp/gemmlab.o && hipcc --offload-arch=gfx950 /tmp/gemmlab.o autogptq_amd/csrc/utils.o -o tools/gemmlab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gemm.hip"
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
__global__ void iota_perm(int* p, int K) {   // a fixed pseudo-random permutation of 0..K-1 (K power-of-two multiple friendly)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) p[i] = (int)(((long long)i * 2731 + 17) % K);
}

int main(int argc, char** argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(init_gemm_device());                        // dynamic LDS above 64 KiB (two-K-group workgroups)
    struct Shape { int M, K, N; };
    std::vector<Shape> shapes = {{2048, 4096, 4096}, {4096, 4096, 4096}, {2048, 4096, 11008}, {2048, 11008, 4096}, {512, 4096, 4096}, {128, 4096, 4096}, {64, 4096, 4096}, {32, 4096, 4096}, {16, 4096, 4096}, {16, 4096, 11008}, {64, 11008, 4096}, {128, 4096, 11008}};
    int only_variant = -1, reps = 5;
    if (argc >= 4) { shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}}; }
    std::vector<int> only_list;
    if (argc >= 5) {
        only_variant = atoi(argv[4]);
        for (const char* c = argv[4]; *c;) { only_list.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
    }
    if (argc >= 6) reps = atoi(argv[5]);
    for (auto s : shapes) {
        const int M = s.M, K = s.K, N = s.N;
        const size_t qw_b = (size_t)K / 8 * N * 4, qz_b = (size_t)(K / 128) * N / 8 * 4, sc_b = (size_t)(K / 128) * N * 2;
        const int nl = 4;   // rotate a few layers (weights are L2/MALL resident in prefill anyway)
        unsigned *qw, *qz; f16 *sc, *x, *out; char* ws; int* perm;
        CK(hipMalloc(&qw, qw_b * nl)); CK(hipMalloc(&qz, qz_b * nl)); CK(hipMalloc(&sc, sc_b * nl));
        CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2)); CK(hipMalloc(&perm, (size_t)K * 4));
        const size_t ws_b = WS_HEADER_BYTES + (size_t)M * K * 2 + (size_t)8 * M * N * 4 + 4096;
        CK(hipMalloc(&ws, ws_b));
        CK(hipMemset(ws, 0, WS_HEADER_BYTES));       // arrival tickets of the in-launch combines start at zero
        fill<<<2048, 256, 0, st>>>(qw, qw_b * nl / 4, 1u);
        fill<<<256, 256, 0, st>>>(qz, qz_b * nl / 4, 2u);
        fill_f16<<<256, 256, 0, st>>>(sc, sc_b * nl / 2, 0.002f, 0.0022f);
        fill_f16<<<2048, 256, 0, st>>>(x, (size_t)M * K, -0.5f, 0.5f);
        iota_perm<<<64, 256, 0, st>>>(perm, K);
        CK(hipStreamSynchronize(st));
        printf("== M=%d K=%d N=%d : %.2f GFLOP per launch\n", M, K, N, 2.0 * M * K * N / 1e9);
        struct V { int id; const char* name; gptq_layer_t L; gptq_tuning_t tu; GemmPlan pl; double us; };
        std::vector<V> vs;
        for (int variant = 0; variant < 26; ++variant) {
            if (only_variant >= 0 && std::find(only_list.begin(), only_list.end(), variant) == only_list.end()) continue;
            V v{}; v.id = variant; v.us = 1e30;
            gptq_layer_t& L = v.L;
            L.K = K; L.N = N; L.bits = 4; L.group_size = 128; L.dtype = GPTQ_F16; L.zero_mode = GPTQ_ZERO_WRAP;
            v.tu.path = 3;
            v.name = "default";
            if (variant == 1) continue;
            if (variant == 5) { if (M < 512) continue; v.tu.reserved[3] = 1; v.name = "VAR0 plain loop"; }
            if (variant == 6) { if (M < 512) continue; v.tu.reserved[3] = 2; v.name = "VAR2 setprio"; }
            if (variant == 7) { if (M < 512) continue; v.tu.reserved[3] = 9; v.name = "ABL -dequant"; }
            if (variant == 8) { if (M < 512) continue; v.tu.reserved[3] = 10; v.name = "ABL -xstaging"; }
            if (variant == 9) { if (M < 512) continue; v.tu.reserved[3] = 11; v.name = "ABL -dequant -xstaging"; }
            if (variant == 10) { if (M < 512) continue; v.tu.reserved[3] = 12; v.name = "ABL -barrier"; }
            if (variant == 11) { if (M < 512) continue; v.tu.reserved[3] = 15; v.name = "ABL -dequant -xstaging -barrier"; }
            if (variant == 3) { if (M > 128) continue; v.tu.reserved[2] = 1; v.name = "forced skinny"; }
            if (variant == 4) { if (M > 128) continue; v.tu.reserved[2] = 2; v.name = "forced tiled"; }
            if (variant == 12) { if (M < 512) continue; L.g_idx = perm; L.perm = perm; L.qweight_seq = qw; v.tu.reserved[3] = 5; v.name = "act-order, register-staged x (no DMA)"; }
            if (variant == 14) { if (M < 512) continue; v.tu.reserved[3] = 3; v.name = "VAR3 cross-step pipeline"; }
            if (variant == 15) { if (M < 512) continue; L.g_idx = perm; L.perm = perm; L.qweight_seq = qw; v.tu.reserved[3] = 3; v.name = "VAR3 cross-step pipeline, act-order + DMA"; }
            if (variant == 16) { if (M < 128) continue; v.tu.reserved[3] = 6; v.name = "KG=1 forced (4 waves)"; }
            if (variant == 17) { if (M < 128) continue; v.tu.reserved[3] = 7; v.name = "KG=2 forced (8 waves, K halves)"; }
            if (variant == 18) { if (M < 128) continue; L.g_idx = perm; L.perm = perm; L.qweight_seq = qw; v.tu.reserved[3] = 6; v.name = "KG=1 forced, act-order + DMA"; }
            if (variant == 19) { if (M < 128) continue; L.g_idx = perm; L.perm = perm; L.qweight_seq = qw; v.tu.reserved[3] = 7; v.name = "KG=2 forced, act-order + DMA"; }
            if (variant == 13) { if (M < 512) continue; L.dtype = GPTQ_BF16; v.name = "bf16 (bit patterns reused: timing only)"; }
            if (variant == 22) { if (M < 512) continue; v.tu.reserved[3] = 24; v.name = "loads interleaved with the MFMA groups (KG = 1)"; }
            if (variant == 23) { if (M < 512) continue; v.tu.reserved[3] = 6; v.name = "default schedule, KG = 1 forced"; }
            if (variant == 24) { if (M < 128) continue; v.tu.reserved[3] = 32; v.name = "ping-pong between the two K groups (KG = 2 forced)"; }
            if (variant == 25) { if (M < 128) continue; L.g_idx = perm; L.perm = perm; L.qweight_seq = qw; v.tu.reserved[3] = 32; v.name = "ping-pong, act-order + DMA (KG = 2 forced)"; }
            if (variant == 20) { if (M < 128) continue; v.tu.reserved[3] = 16; v.name = "timeline (s_memtime stamps), one K group"; }
            if (variant == 21) { if (M < 128) continue; v.tu.reserved[3] = 17; v.name = "timeline (s_memtime stamps), two K groups"; }
            if (variant == 2) { L.g_idx = perm; L.perm = perm; L.qweight_seq = qw; v.name = "act-order (x permute + qweight_seq)"; }
            v.pl = plan_gemm(L, M, &v.tu);
            if (!v.pl.supported) { printf("  unsupported\n"); continue; }
            vs.push_back(v);
        }
        auto launch_all = [&](V& v) {
            for (int i = 0; i < nl; ++i) {
                gptq_layer_t L = v.L;
                L.qweight = qw + (size_t)i * qw_b / 4; L.qzeros = qz + (size_t)i * qz_b / 4; L.scales = sc + (size_t)i * sc_b / 2;
                L.qweight_seq = (v.id == 2 || v.id == 12 || v.id == 15 || v.id == 18 || v.id == 19 || v.id == 25) ? L.qweight : nullptr;
                hipError_t e = launch_gemm(L, v.pl, x, out, M, ws, ws + WS_HEADER_BYTES, st);
                if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
            }
        };
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 20; ++w) for (auto& v : vs) launch_all(v);          // warm the clocks up
        CK(hipStreamSynchronize(st));
        for (int round = 0; round < 5; ++round)                                     // interleaved rounds, min per variant
            for (auto& v : vs) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) launch_all(v);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                v.us = std::min(v.us, ms * 1e3 / (reps * nl));
            }
        for (auto& v : vs)
            printf("  %9.2f us  %8.1f TFLOP/s  %7.1f GB/s(w)  %-38s %s mt=%d bk=%d grid=%dx%d ksplit=%d\n", v.us, 2.0 * M * K * N / v.us / 1e6, qw_b / v.us / 1e3, v.name,
                   v.pl.skinny ? "skinny" : "tiled", v.pl.mt, v.pl.bk, v.pl.nbm, v.pl.nbn, v.pl.ksplit);
        for (auto& v : vs) {
            if (v.id != 20 && v.id != 21) continue;
            // one more launch with a clean stamp area: per wave of workgroup 0 the cycle sums of the five phases of a K-step
            CK(hipMemsetAsync(ws + WS_HEADER_BYTES, 0, 16 * 8 * 4, st));
            gptq_layer_t L = v.L; L.qweight = qw; L.qzeros = qz; L.scales = sc;
            if (launch_gemm(L, v.pl, x, out, M, ws, ws + WS_HEADER_BYTES, st) != hipSuccess) { printf("timeline launch failed\n"); continue; }
            CK(hipStreamSynchronize(st));
            unsigned hst[16 * 8];
            CK(hipMemcpy(hst, ws + WS_HEADER_BYTES, sizeof(hst), hipMemcpyDeviceToHost));
            printf("  %s: shader cycles per K-step, workgroup 0 (issue loads | first fragments ready | MFMA groups issued | x tile stored | barrier released)\n", v.name);
            for (int w = 0; w < (v.pl.kg == 2 ? 8 : 4); ++w) {
                const unsigned* o = hst + w * 8;
                const double n = o[5] ? (double)o[5] : 1.0;
                printf("    wave %d (%u steps): %7.0f | %7.0f | %7.0f | %7.0f | %7.0f   = %7.0f cycles per step\n", w, o[5], o[0] / n, o[1] / n, o[2] / n, o[3] / n, o[4] / n,
                       (o[0] + o[1] + o[2] + o[3] + o[4]) / n);
            }
        }
        CK(hipFree(qw)); CK(hipFree(qz)); CK(hipFree(sc)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(ws)); CK(hipFree(perm));
    }
    return 0;
}
