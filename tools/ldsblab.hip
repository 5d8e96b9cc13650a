// ldsblab.hip -- the LDS-shared-B prefill kernel (csrc/gemm_ldsb.hip) against the product's tiled kernel (through the C ABI): same inputs, output
// comparison, interleaved timing.   (measurement tool, not product)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I include -I autogptq_amd/csrc tools/ldsblab.hip -o tools/ldsblab -L autogptq_amd -lgptq_mi355x -Wl,-rpath,'$ORIGIN/../autogptq_amd'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#define GPTQ_LDSB_ABLATIONS 1
#include "gemm_ldsb.hip"
using namespace gptq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { printf("gptq error %d (%s) at %s:%d\n", r_, gptq_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_u32(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7) ^ seed;
        v ^= v << 13; v ^= v >> 17; v ^= v << 5;
        p[i] = v;
    }
}
__global__ void fill_f16(f16* p, size_t n, float lo, float hi, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)(i * 2246822519u) ^ seed; v ^= v >> 15; v *= 2654435761u; v ^= v >> 13;
        p[i] = (f16)(lo + (hi - lo) * (float)(v & 0xffff) / 65536.f);
    }
}
__global__ void max_diff(const f16* a, const f16* b, size_t n, float* out) {   // out[0] = max |a - b|, out[1] = max |b|
    float d = 0.f, s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d = fmaxf(d, fabsf((float)a[i] - (float)b[i])); s = fmaxf(s, fabsf((float)b[i]));
    }
    atomicMax((int*)out, __float_as_int(d)); atomicMax((int*)out + 1, __float_as_int(s));
}

int main(int argc, char** argv) {
    struct Shape { int M, K, N; };
    std::vector<Shape> shapes = {{2048, 4096, 4096}, {4096, 4096, 4096}, {2048, 4096, 11008}, {2048, 11008, 4096}, {300, 1024, 640}};
    if (argc >= 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
    const int reps = argc >= 5 ? atoi(argv[4]) : 5;
    GK(gptq_init());
    CK(init_gemm_ldsb_device());
    hipStream_t st; CK(hipStreamCreate(&st));
    for (auto s : shapes) {
        const int M = s.M, K = s.K, N = s.N, gs = 128, nl = 4;
        const size_t qw_b = (size_t)K / 8 * N * 4, qz_b = (size_t)(K / gs) * N / 8 * 4, sc_b = (size_t)(K / gs) * N * 2;
        unsigned *qw, *qz; f16 *sc, *x, *o1, *o2; char* ws; float* d;
        CK(hipMalloc(&qw, qw_b * nl)); CK(hipMalloc(&qz, qz_b * nl)); CK(hipMalloc(&sc, sc_b * nl));
        CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&o1, (size_t)M * N * 2)); CK(hipMalloc(&o2, (size_t)M * N * 2)); CK(hipMalloc(&d, 8));
        fill_u32<<<2048, 256, 0, st>>>(qw, qw_b * nl / 4, 1u);
        fill_u32<<<256, 256, 0, st>>>(qz, qz_b * nl / 4, 2u);
        fill_f16<<<256, 256, 0, st>>>(sc, sc_b * nl / 2, 0.002f, 0.0022f, 3u);
        fill_f16<<<2048, 256, 0, st>>>(x, (size_t)M * K, -0.5f, 0.5f, 4u);
        std::vector<gptq_layer_t> Ls(nl);
        size_t wsb = 1 << 20;
        for (int i = 0; i < nl; ++i) {
            gptq_layer_t L{}; L.qweight = qw + i * qw_b / 4; L.qzeros = qz + i * qz_b / 4; L.scales = sc + i * sc_b / 2;
            L.K = K; L.N = N; L.bits = 4; L.group_size = gs; L.dtype = GPTQ_F16; L.zero_mode = GPTQ_ZERO_WRAP;
            Ls[i] = L;
            wsb = std::max(wsb, gptq_workspace_bytes(&L, M));
        }
        CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
        printf("== M=%d K=%d N=%d : %.2f GFLOP\n", M, K, N, 2.0 * M * K * N / 1e9);
        if (!ldsb_supported(Ls[0], M)) { printf("  ldsb: unsupported\n"); continue; }
        // correctness
        CK(hipMemsetAsync(o1, 0xFF, (size_t)M * N * 2, st)); CK(hipMemsetAsync(d, 0, 8, st));
        GK(gptq_forward(&Ls[1], x, o2, M, ws, wsb, st));
        struct Cfg { int bk, kg; const char* name; };
        std::vector<Cfg> cfgs = {{64, 1, "BK=64, 1 group (96 KiB LDS: 1 workgroup/CU)"}, {32, 1, "BK=32, 1 group (48 KiB: 2 workgroups/CU)"}};
        if ((K / 32) % 4 == 0) cfgs.push_back({32, 2, "BK=32, 2 K groups (8 waves)"});
        for (auto& c : cfgs) {
            CK(hipMemsetAsync(o1, 0xFF, (size_t)M * N * 2, st)); CK(hipMemsetAsync(d, 0, 8, st));
            CK(launch_gemm_ldsb(Ls[1], Ls[1].qweight, x, o1, M, st, c.bk, c.kg));
            max_diff<<<1024, 256, 0, st>>>(o1, o2, (size_t)M * N, d);
            float hd[2]; CK(hipMemcpyAsync(hd, d, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            printf("  %-46s max |ldsb - product| = %.4e (output scale %.3e) => %s\n", c.name, hd[0], hd[1], (hd[0] <= 2e-3f * hd[1]) ? "PASS" : "FAIL");
        }
        // timing (interleaved rounds, min)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<double> best(cfgs.size() + 1, 1e30);
        for (int w = 0; w < 5; ++w) for (int i = 0; i < nl; ++i) { for (auto& c : cfgs) CK(launch_gemm_ldsb(Ls[i], Ls[i].qweight, x, o1, M, st, c.bk, c.kg)); GK(gptq_forward(&Ls[i], x, o2, M, ws, wsb, st)); }
        for (int round = 0; round < 5; ++round)
            for (size_t v = 0; v <= cfgs.size(); ++v) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r)
                    for (int i = 0; i < nl; ++i) {
                        if (v < cfgs.size()) CK(launch_gemm_ldsb(Ls[i], Ls[i].qweight, x, o1, M, st, cfgs[v].bk, cfgs[v].kg));
                        else GK(gptq_forward(&Ls[i], x, o2, M, ws, wsb, st));
                    }
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best[v] = std::min(best[v], ms * 1e3 / (reps * nl));
            }
        for (size_t v = 0; v < cfgs.size(); ++v) printf("  %9.2f us  %8.1f TFLOP/s  ldsb %s\n", best[v], 2.0 * M * K * N / best[v] / 1e6, cfgs[v].name);
        printf("  %9.2f us  %8.1f TFLOP/s  product tiled kernel (gptq_forward)\n", best[cfgs.size()], 2.0 * M * K * N / best[cfgs.size()] / 1e6);
        if (M >= 2048) {
            static const int abls[] = {1, 2, 4, 6, 8, 16, 31};
            static const char* an[] = {"-dequant math", "-x DMA", "-barrier", "-x DMA -barrier", "-weight loads", "-B store", "-everything but MFMA + LDS reads"};
            for (int ai = 0; ai < 7; ++ai) {
                double b = 1e30;
                for (int round = 0; round < 3; ++round) {
                    CK(hipEventRecord(e0, st));
                    for (int r = 0; r < reps; ++r) for (int i = 0; i < nl; ++i) CK(launch_gemm_ldsb(Ls[i], Ls[i].qweight, x, o1, M, st, 32, 1, abls[ai]));
                    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    b = std::min(b, ms * 1e3 / (reps * nl));
                }
                printf("  %9.2f us  %8.1f TFLOP/s    ablation (BK=32, 1 group) %s\n", b, 2.0 * M * K * N / b / 1e6, an[ai]);
            }
        }
        CK(hipFree(qw)); CK(hipFree(qz)); CK(hipFree(sc)); CK(hipFree(x)); CK(hipFree(o1)); CK(hipFree(o2)); CK(hipFree(ws)); CK(hipFree(d));
    }
    return 0;
}
