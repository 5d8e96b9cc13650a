// LAB: what bounds the x path of the 17..128-row kernel (gemm_mid.hip)?  Every workgroup reads the SAME L2-resident x[M][K] (fp16) once, the way a
// 64-column strip's waves do, in different segment shapes and through the two paths, and nothing else:
//   seg 64 :  one instruction = 16 rows x  64 contiguous bytes (a 32-deep K-step of a 16-row tile)   -- what the kernels do today
//   seg 128:  one instruction =  8 rows x 128 contiguous bytes (two K-steps of half a row tile: whole 128-byte lines)
//   seg 256:  one instruction =  4 rows x 256 bytes
//   path dma: global_load_lds_dwordx4 (1 KiB into LDS)    path reg: global_load_dwordx4 into registers (xor-reduced so it is not dead)
// Reported: us per launch and GB/s per CU for 256 workgroups of 8 waves.   hipcc --offload-arch=gfx950 -O2 tools/xfetch_lab.hip -o /tmp/xfetch_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// SEG = bytes of one row segment per instruction (64 / 128 / 256); rows per instruction = 1024 / SEG
// ROT: every workgroup starts its walk along K at a different segment (blockIdx * 37 mod segments per wave) -- do 256 workgroups reading the SAME
// addresses at the same moment hot-spot the L2 channels?
template <int SEG, bool DMA, bool ROT = false>
__global__ void __launch_bounds__(512) xfetch_kernel(const char* x, int M, int K, unsigned* sink, int depth) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;                 // lanes per row, rows per instruction
    const int r = lane / LPR, o = lane % LPR;
    const size_t rowb = (size_t)K * 2;
    const int segs = (int)(rowb / SEG);                           // segments along K
    const int spw = (segs + W - 1) / W;
    const int s0 = wave * spw, s1 = min(s0 + spw, segs);
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + wave * 16384));
    u32x4 acc = {0, 0, 0, 0};
    int slot = 0;
    const int cnt = s1 - s0;
    const int rot = (ROT && cnt > 0) ? (int)((blockIdx.x * 37u) % (unsigned)cnt) : 0;
    for (int si = 0; si < cnt; ++si) {
        int s = s0 + si + rot;
        if (s >= s1) s -= cnt;
        for (int rb = 0; rb < M; rb += RPI) {
            const char* src = x + (size_t)min(rb + r, M - 1) * rowb + (size_t)s * SEG + o * 16;
            if constexpr (DMA) {
                dma16(src, __builtin_amdgcn_readfirstlane(lds + (slot & 15) * 1024));
                if ((++slot % depth) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                const u32x4 v = *(const u32x4*)src;
                acc ^= v;
            }
        }
    }
    if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *(const u32x4*)(smem + wave * 16384 + lane * 16);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = acc[0];
}

template <int SEG, bool DMA, bool ROT = false>
static void run(const char* name, const char* x, int M, int K, unsigned* sink, int depth) {
    CK(hipFuncSetAttribute((const void*)xfetch_kernel<SEG, DMA, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((xfetch_kernel<SEG, DMA, ROT>), dim3(256), dim3(512), 128 * 1024, 0, x, M, K, sink, depth);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((xfetch_kernel<SEG, DMA, ROT>), dim3(256), dim3(512), 128 * 1024, 0, x, M, K, sink, depth);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)M * K * 2;
    printf("  %-34s %7.2f us per launch   %6.1f GB/s per CU   (%.0f KB per workgroup)\n", name, us, bytes / (us * 1e-6) / 1e9, bytes / 1024);
}

int main() {
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    for (int M : {64, 128}) for (int K : {4096, 11008}) {
        char* x; CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMemset(x, 1, (size_t)M * K * 2));
        printf("== x[%d][%d] fp16 read once by each of 256 workgroups x 8 waves\n", M, K);
        run<64, true>("dma, 16 rows x 64 B, depth 8", x, M, K, sink, 8);
        run<128, true>("dma,  8 rows x 128 B, depth 8", x, M, K, sink, 8);
        run<256, true>("dma,  4 rows x 256 B, depth 8", x, M, K, sink, 8);
        run<64, true>("dma, 16 rows x 64 B, depth 16", x, M, K, sink, 16);
        run<128, true>("dma,  8 rows x 128 B, depth 16", x, M, K, sink, 16);
        run<64, false>("reg, 16 rows x 64 B", x, M, K, sink, 0);
        run<128, false>("reg,  8 rows x 128 B", x, M, K, sink, 0);
        run<256, false>("reg,  4 rows x 256 B", x, M, K, sink, 0);
        run<64, true, true>("dma, 16 rows x 64 B, depth 8, ROTATED", x, M, K, sink, 8);
        run<128, true, true>("dma,  8 rows x 128 B, depth 16, ROTATED", x, M, K, sink, 16);
        run<128, false, true>("reg,  8 rows x 128 B, ROTATED", x, M, K, sink, 0);
        CK(hipFree(x));
    }
    return 0;
}
