// Lab microbenchmark (no product path): how fast can ONE workgroup per CU pull an L2-resident buffer that every workgroup reads (the x of a batched-decode
// launch)?  256 workgroups each read the same `bytes` of global memory: (a) dwordx4 loads into registers, (b) global_load_lds_dwordx4 into the LDS,
// for 4 / 8 / 16 waves per workgroup and 2 / 4 / 8 loads in flight per wave.  Prints GB/s per CU and the aggregate.
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/xpull.hip -o tools/lab/xpull ; run on the GPU box: tools/lab/xpull
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(1024) pull_regs(const u32x4* __restrict__ x, size_t n16, unsigned* sink, int reps) {
    const int tid = threadIdx.x, nt = blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = tid; i + (size_t)(U - 1) * nt < n16; i += (size_t)U * nt) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(x + i + (size_t)u * nt);
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
        asm volatile("" : "+v"(acc));
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = 1;
}

template <int U>
__global__ void __launch_bounds__(1024) pull_regs_plain(const u32x4* __restrict__ x, size_t n16, unsigned* sink, int reps) {
    const int tid = threadIdx.x, nt = blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = tid; i + (size_t)(U - 1) * nt < n16; i += (size_t)U * nt) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = x[i + (size_t)u * nt];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
        asm volatile("" : "+v"(acc));
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = 1;
}

// LDS DMA: each wave fills its own 1 KiB x U slots round-robin (the data is never read: the pull rate is what is measured)
template <int U>
__global__ void __launch_bounds__(1024) pull_lds(const char* __restrict__ x, size_t bytes, unsigned* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem) + wave * U * 1024;
    for (int r = 0; r < reps; ++r) {
        for (size_t off = (size_t)wave * 1024; off + (size_t)(U - 1) * nw * 1024 < bytes; off += (size_t)U * nw * 1024) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const char* src = x + off + (size_t)u * nw * 1024;
                const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + u * 1024);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"((unsigned)lane * 16u), "s"(src) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (smem[tid] == 0x7f && reps < 0) sink[blockIdx.x] = 1;
}

int main() {
    const int WGS = 256, reps = 8;
    unsigned* sink;
    hipMalloc(&sink, WGS * 4);
    for (size_t kb : {64, 128, 256, 512, 1024}) {
        const size_t bytes = kb * 1024;
        char* x;
        hipMalloc(&x, bytes);
        hipMemset(x, 1, bytes);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        auto time_it = [&](auto launch, const char* name, int waves, int u) {
            launch(); launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / 20.0, per_cu = (double)bytes * reps / (us * 1e-6) / 1e9;
            printf("%5zu KiB  %-12s waves=%2d U=%d  %8.2f us per launch (%d passes)  %7.1f GB/s per CU  %6.2f TB/s chip\n", kb, name, waves, u, us, reps, per_cu, per_cu * WGS / 1e3);
        };
        for (int waves : {4, 8, 16}) {
            const dim3 g(WGS), b(waves * 64);
            time_it([&] { hipLaunchKernelGGL(pull_regs_plain<4>, g, b, 0, 0, (const u32x4*)x, bytes / 16, sink, reps); }, "regs", waves, 4);
            time_it([&] { hipLaunchKernelGGL(pull_regs_plain<8>, g, b, 0, 0, (const u32x4*)x, bytes / 16, sink, reps); }, "regs", waves, 8);
            time_it([&] { hipLaunchKernelGGL(pull_regs<8>, g, b, 0, 0, (const u32x4*)x, bytes / 16, sink, reps); }, "regs-nt", waves, 8);
            time_it([&] { hipLaunchKernelGGL(pull_lds<4>, g, b, waves * 4 * 1024, 0, (const char*)x, bytes, sink, reps); }, "lds-dma", waves, 4);
            time_it([&] { hipLaunchKernelGGL(pull_lds<8>, g, b, waves * 8 * 1024, 0, (const char*)x, bytes, sink, reps); }, "lds-dma", waves, 8);
        }
        hipFree(x);
    }
    return 0;
}
