// membench2.hip -- round 4: does a STRIP-MAJOR side copy of the packed weights stream faster than the checkpoint's row-major layout,
// at the same workgroup count and with no K split?  ("Next round" item 2 (i).)
// Not part of the product; build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/membench2.hip -o tools/membench2
//
// Each launch reads one packed 4-bit matrix [R = K/8 rows][N words] exactly once with the decomposition of the decode kernels: a workgroup owns
// a strip of CT = 4*LN columns over ALL rows; a lane owns 16 bytes of a packed row; the 64/LN row slots of a wave x W waves x U instructions in
// flight walk the rows.  Two source layouts over the same bytes:
//   ROWMAJ   word (r, n) at r * N + n                        -- a wave instruction = 64/LN row segments of 16*LN bytes, N*4 bytes apart
//   STRIPMAJ word (r, n) at (n / CT) * R * CT + r * CT + n % CT -- a strip is one contiguous run; a wave instruction = one contiguous 1 KiB
// and two load paths: registers (global_load_dwordx4 nt) or LDS DMA (global_load_lds_dwordx4 nt into the wave's landing area, read back).
// Launches are captured in a hipGraph over a rotating set of matrices (> 256 MiB: HBM-cold) or over ONE matrix (cache-hot).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ void dma16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// LAYOUT 0 = row-major with the PRODUCT's row assignment (a lane's U rows are consecutive), 1 = strip-major (an instruction's rows are consecutive),
// 2 = row-major with strip-major's row assignment (isolates the layout from the assignment).  MODE 0 = registers, 1 = LDS DMA.
template <int LN, int U, int LAYOUT, int MODE>
__global__ void __launch_bounds__(1024) reader(const unsigned* __restrict__ q, int rows, int N, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WR = 64 / LN, CT = LN * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const size_t sbase = (LAYOUT == 1) ? (size_t)strip * rows * CT : (size_t)strip * CT;
    const size_t rstride = (LAYOUT == 1) ? CT : N;
    char* const wq = smem + (size_t)wave * (U * 1024);
    const unsigned wq_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)wq);
    unsigned acc = 0;
    const int rows_per_iter = W * WR * U;
    for (int base = 0; base < rows; base += rows_per_iter) {
        u32x4 v[U];
        if (MODE == 1 && base) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < U; ++j) {
            int r = (LAYOUT == 0) ? base + (wave * WR + rs) * U + j : base + (wave * U + j) * WR + rs;
            r = min(r, rows - 1);
            const unsigned* p = q + sbase + (size_t)r * rstride + cl * 4;
            if (MODE == 0) v[j] = __builtin_nontemporal_load((const u32x4*)p);
            else dma16_nt(p, wq_lds + j * 1024);
        }
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = *(const u32x4*)(wq + j * 1024 + lane * 16);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// Reference: the same bytes as one flat contiguous stream, grid-stride over 1 KiB wave chunks (no strip structure at all).
template <int U>
__global__ void __launch_bounds__(1024) flat_reader(const unsigned* __restrict__ q, size_t words, unsigned* out) {
    const size_t nchunk = words / 256;                           // 1 KiB chunks
    const size_t per = (nchunk + gridDim.x - 1) / gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const size_t c0 = (size_t)blockIdx.x * per, c1 = min(c0 + per, nchunk);
    unsigned acc = 0;
    for (size_t c = c0 + wave * U; c < c1; c += (size_t)W * U) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = __builtin_nontemporal_load((const u32x4*)(q + min(c + j, c1 - 1) * 256 + lane * 4));
#pragma unroll
        for (int j = 0; j < U; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

struct Res { float us; char name[128]; };

template <typename F>
static float time_graph(F launch_one, size_t mats, int reps, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (size_t i = 0; i < mats; ++i) launch_one(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1e3f / mats);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

template <int LN, int U, int LAYOUT, int MODE>
static void run(std::vector<Res>& res, const unsigned* buf, size_t mats, size_t launches, int rows, int N, int W, unsigned* out, hipStream_t st) {
    if (N % (LN * 4)) return;
    const int strips = N / (LN * 4);
    const size_t mat_words = (size_t)rows * N;
    const size_t lds = MODE == 1 ? (size_t)W * U * 1024 : 0;
    if (lds > 160 * 1024) return;
    static bool granted = false;
    if (lds > 65536 && !granted) { CK(hipFuncSetAttribute((const void*)reader<LN, U, LAYOUT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); granted = true; }
    auto one = [&](size_t i) { reader<LN, U, LAYOUT, MODE><<<dim3(strips), dim3(W * 64), lds, st>>>(buf + (i % mats) * mat_words, rows, N, out); };
    Res r;
    r.us = time_graph(one, launches, 5, st);
    snprintf(r.name, sizeof r.name, "%s %s LN=%2d (%3dB seg) W=%2d U=%d  wgs=%4d", LAYOUT == 1 ? "STRIPMAJ" : (LAYOUT == 0 ? "rowmaj  " : "rowmaj* "), MODE ? "dma" : "reg",
             LN, LN * 16, W, U, strips);
    res.push_back(r);
}

template <int LN, int U>
static void run4(std::vector<Res>& res, const unsigned* buf, size_t mats, size_t launches, int rows, int N, int W, unsigned* out, hipStream_t st) {
    run<LN, U, 0, 0>(res, buf, mats, launches, rows, N, W, out, st);
    run<LN, U, 1, 0>(res, buf, mats, launches, rows, N, W, out, st);
    run<LN, U, 2, 0>(res, buf, mats, launches, rows, N, W, out, st);
    run<LN, U, 0, 1>(res, buf, mats, launches, rows, N, W, out, st);
    run<LN, U, 1, 1>(res, buf, mats, launches, rows, N, W, out, st);
}

__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
}

int main(int argc, char** argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t total = (size_t)1024 << 20;
    unsigned* buf; CK(hipMalloc(&buf, total));
    unsigned* out; CK(hipMalloc(&out, 64));
    fill<<<2048, 256, 0, st>>>(buf, total / 4);
    CK(hipStreamSynchronize(st));
    struct Shape { int K, N; const char* what; } shapes[] = {{4096, 4096, "o / q / k / v"}, {11008, 4096, "down"}, {4096, 11008, "gate / up"},
                                                            {4096, 12288, "q|k|v as one launch"}, {4096, 22016, "gate|up as one launch"}};
    for (int hot = 0; hot < 2; ++hot)
        for (auto s : shapes) {
            const int rows = s.K / 8, N = s.N;
            const size_t bytes = (size_t)rows * N * 4;
            const size_t mats = hot ? 1 : total / bytes;
            const size_t launches = std::max<size_t>(mats, 24);
            printf("== K=%d N=%d (%s): %zu B per launch, %s (%zu matrices)\n", s.K, s.N, s.what, bytes, hot ? "cache-HOT: one matrix replayed" : "HBM-COLD rotation", mats);
            std::vector<Res> res;
            for (int W : {4, 8, 16}) {
                run4<4, 2>(res, buf, mats, launches, rows, N, W, out, st);
                run4<4, 4>(res, buf, mats, launches, rows, N, W, out, st);
                run4<4, 8>(res, buf, mats, launches, rows, N, W, out, st);
                run4<8, 2>(res, buf, mats, launches, rows, N, W, out, st);
                run4<8, 4>(res, buf, mats, launches, rows, N, W, out, st);
                run4<8, 8>(res, buf, mats, launches, rows, N, W, out, st);
                run4<16, 4>(res, buf, mats, launches, rows, N, W, out, st);
                run4<16, 8>(res, buf, mats, launches, rows, N, W, out, st);
            }
            for (int W : {1, 2}) {                                 // very small workgroups only make sense with contiguous strips
                run4<4, 8>(res, buf, mats, launches, rows, N, W, out, st);
                run4<2, 8>(res, buf, mats, launches, rows, N, W, out, st);
                run4<2, 4>(res, buf, mats, launches, rows, N, W, out, st);
            }
            for (int grid : {256, 512, 1024, 2048}) {
                const size_t words = bytes / 4;
                auto one = [&](size_t i) { flat_reader<4><<<dim3(grid), dim3(256), 0, st>>>(buf + (i % mats) * words, words, out); };
                Res r; r.us = time_graph(one, launches, 5, st);
                snprintf(r.name, sizeof r.name, "FLAT contiguous stream, grid=%d x 256 thr, U=4", grid);
                res.push_back(r);
            }
            std::sort(res.begin(), res.end(), [](const Res& a, const Res& b) { return a.us < b.us; });
            for (size_t i = 0; i < res.size(); ++i)
                if (i < 40 || strstr(res[i].name, "W=16 U=2") || strstr(res[i].name, "FLAT"))
                    printf("  %8.2f us  %7.1f GB/s  %s\n", res[i].us, bytes / res[i].us / 1e3, res[i].name);
            // the head-to-head the question is about: per (LN, W, U, path) strip-major against row-major
            printf("  -- pairs (rowmaj -> STRIPMAJ), same geometry:\n");
            for (auto& a : res) {
                if (strncmp(a.name, "rowmaj  ", 8)) continue;
                for (auto& b : res)
                    if (!strncmp(b.name, "STRIPMAJ", 8) && !strcmp(a.name + 8, b.name + 8))
                        printf("     %s : %6.2f -> %6.2f us  (%+.1f %%)\n", a.name + 9, a.us, b.us, (a.us / b.us - 1.f) * 100.f);
            }
            fflush(stdout);
        }
    return 0;
}
