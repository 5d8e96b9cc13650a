// membench.hip -- one-shot streaming-read floor for GEMV-sized transfers on MI355X.
// Not part of the product; a measurement tool (build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench).
// Each "launch" reads one [R rows x N*4 bytes] int32 matrix (a packed 4-bit weight) exactly once, from a
// rotating set of matrices > 256 MiB (Infinity Cache cannot serve them), with the access pattern of a
// GEMV strip decomposition:  block = (strip of CW columns) x (row range);  lane = 16 B.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LN lanes of a wave along columns (LN*4 columns = LN*16 bytes contiguous), 64/LN row slots per wave.
// Block = W waves; block (bx, by) owns strip bx and rows [by*rps, (by+1)*rps).  U loads in flight per lane.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
template <int LN, int U, bool NT, bool REMAP = false>
__global__ void __launch_bounds__(1024) strip_read(const unsigned* __restrict__ q, int rows, int N, int rps, unsigned* out) {
    constexpr int WR = 64 / LN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int n0 = (strip * LN + cl) * 4;
    const int rb = blockIdx.y * rps, re = min(rb + rps, rows);
    unsigned acc = 0;
    const int step = W * WR;
    for (int base = rb; base < re; base += U * step) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            int r = base + j * step + wave * WR + rs;
            r = min(r, re - 1);
            const u32x4* p = (const u32x4*)(q + (size_t)r * N + n0);
            v[j] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    if (acc == 0x12345678u) out[0] = acc;   // never true for random data; keeps the loads alive
}

template <int LN, int U, bool NT, bool REMAP = false>
float run(const unsigned* buf, size_t mats, int rows, int N, int waves, int ksplit, unsigned* out, int reps, hipStream_t st) {
    const int strips = N / (LN * 4);
    const int rps = (rows + ksplit - 1) / ksplit;
    dim3 grid(strips, ksplit), block(waves * 64);
    const size_t mat_words = (size_t)rows * N;
    for (size_t i = 0; i < mats; ++i) strip_read<LN, U, NT, REMAP><<<grid, block, 0, st>>>(buf + i * mat_words, rows, N, rps, out);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        for (size_t i = 0; i < mats; ++i) strip_read<LN, U, NT, REMAP><<<grid, block, 0, st>>>(buf + i * mat_words, rows, N, rps, out);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * mats);
}

__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
}

int main(int argc, char** argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t total = (size_t)768 << 20;
    unsigned* buf; CK(hipMalloc(&buf, total));
    unsigned* out; CK(hipMalloc(&out, 64));
    fill<<<2048, 256, 0, st>>>(buf, total / 4);
    CK(hipStreamSynchronize(st));
    struct Shape { int K, N; } shapes[] = {{4096, 4096}, {4096, 11008}, {11008, 4096}};
    for (auto s : shapes) {
        const int rows = s.K / 8, N = s.N;
        const size_t bytes = (size_t)rows * N * 4;
        const size_t mats = total / bytes;
        printf("== K=%d N=%d : %zu B per launch, %zu rotating matrices\n", s.K, s.N, bytes, mats);
        struct R { float us; char name[96]; };
        std::vector<R> res;
#define TRY(LN, U, NT, W, KS) do { if (N % (LN * 4) == 0) { R r; r.us = run<LN, U, NT>(buf, mats, rows, N, W, KS, out, 3, st); \
        snprintf(r.name, sizeof r.name, "LN=%2d U=%d nt=%d waves=%2d ksplit=%3d blocks=%5d", LN, U, NT, W, KS, N / (LN * 4) * KS); res.push_back(r); } } while (0)
#define TRYR(LN, U, W, KS) do { if (N % (LN * 4) == 0) { R r; r.us = run<LN, U, true, true>(buf, mats, rows, N, W, KS, out, 3, st); \
        snprintf(r.name, sizeof r.name, "LN=%2d U=%d nt=1 waves=%2d ksplit=%3d blocks=%5d XCD-REMAP", LN, U, W, KS, N / (LN * 4) * KS); res.push_back(r); \
        r.us = run<LN, U, false, true>(buf, mats, rows, N, W, KS, out, 3, st); \
        snprintf(r.name, sizeof r.name, "LN=%2d U=%d nt=0 waves=%2d ksplit=%3d blocks=%5d XCD-REMAP", LN, U, W, KS, N / (LN * 4) * KS); res.push_back(r); } } while (0)
        for (int W : {4, 8, 16}) {
            TRYR(4, 4, W, 1); TRYR(4, 8, W, 1); TRYR(4, 2, W, 1); TRYR(8, 4, W, 1); TRYR(8, 8, W, 1); TRYR(8, 4, W, 2); TRYR(16, 4, W, 1); TRYR(16, 4, W, 4);
            TRY(16, 4, true, W, 4); TRY(64, 1, true, W, 64);
        }
        std::sort(res.begin(), res.end(), [](const R& a, const R& b) { return a.us < b.us; });
        for (size_t i = 0; i < res.size(); ++i)
            if (true) printf("  %8.2f us  %7.1f GB/s  %s\n", res[i].us, bytes / res[i].us / 1e3, res[i].name);
    }
    return 0;
}
