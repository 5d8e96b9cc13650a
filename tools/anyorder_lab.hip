// LAB: does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) clear the AQL barrier bit on gfx950 / ROCm 7.2, i.e. may the next kernel of the SAME
// stream start before the previous one has finished -- and are the workgroups of the two still dispatched in packet order?  (The chained decode
// launches of DESIGN 4.1d rest on both.)  Standalone: hipcc --offload-arch=gfx950 -O2 tools/anyorder_lab.hip -o gpurun_out/anyorder_lab
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", e_, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long* out, int base, long long ticks) {
    unsigned long long t0 = wall_clock64();
    unsigned long long t1 = t0;
    for (int i = 0; i < 4000000 && (long long)(t1 - t0) < ticks; ++i) { __builtin_amdgcn_s_sleep(8); t1 = wall_clock64(); }
    if (threadIdx.x == 0) { out[2 * (base + blockIdx.x)] = t0; out[2 * (base + blockIdx.x) + 1] = t1; }
}

static void report(const char* name, const std::vector<unsigned long long>& h, int nA, int nB) {
    unsigned long long a0 = ~0ull, a0max = 0, a1 = 0, b0 = ~0ull, b1 = 0;
    for (int i = 0; i < nA; ++i) { a0 = std::min(a0, h[2 * i]); a0max = std::max(a0max, h[2 * i]); a1 = std::max(a1, h[2 * i + 1]); }
    for (int i = nA; i < nA + nB; ++i) { b0 = std::min(b0, h[2 * i]); b1 = std::max(b1, h[2 * i + 1]); }
    // wall_clock64 = 100 MHz
    printf("%-44s A: first start 0, last start %+7.2f us, end %+7.2f us | B: first start %+7.2f us, end %+7.2f us  => %s, B starts %s the last A workgroup started\n", name,
           (a0max - a0) / 100.0, (a1 - a0) / 100.0, ((long long)b0 - (long long)a0) / 100.0, ((long long)b1 - (long long)a0) / 100.0,
           b0 < a1 ? "OVERLAP" : "serial", b0 >= a0max ? "after" : "BEFORE");
}

int main() {
    int nA = 1, nB = 1;
    unsigned long long* d; CK(hipMalloc(&d, 2 * 8 * 16384));
    std::vector<unsigned long long> h(2 * 16384);
    hipStream_t st; CK(hipStreamCreate(&st));
    auto fetch = [&]() { CK(hipStreamSynchronize(st)); CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost)); };
    for (int geo = 0; geo < 2; ++geo) {
        nA = geo ? 4096 : 1; nB = geo ? 256 : 1;
        long long tA = geo ? 500 : 5000, tB = 100;       // 5 us per workgroup x 4096 workgroups, or one 50 us workgroup
        int thr = 256;
        printf("--- A = %d workgroups x %d threads spinning %.0f us, B = %d workgroups\n", nA, thr, tA / 100.0, nB);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(spin_kernel, dim3(nA), dim3(thr), 0, st, d, 0, tA);
            hipLaunchKernelGGL(spin_kernel, dim3(nB), dim3(thr), 0, st, d, nA, tB);
            fetch(); if (rep) report("plain launches", h, nA, nB);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipExtLaunchKernelGGL(spin_kernel, dim3(nA), dim3(thr), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, 0, tA);
            hipExtLaunchKernelGGL(spin_kernel, dim3(nB), dim3(thr), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, nA, tB);
            fetch(); if (rep) report("hipExtLaunchKernel any-order", h, nA, nB);
        }
        {   // stream capture of the any-order pair
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            hipExtLaunchKernelGGL(spin_kernel, dim3(nA), dim3(thr), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, 0, tA);
            hipExtLaunchKernelGGL(spin_kernel, dim3(nB), dim3(thr), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, nA, tB);
            hipError_t e = hipStreamEndCapture(st, &g);
            if (e == hipSuccess && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                for (int rep = 0; rep < 2; ++rep) { CK(hipGraphLaunch(ge, st)); fetch(); }
                report("captured any-order pair (graph replay)", h, nA, nB);
            } else printf("capture of any-order launches failed: %d\n", e);
        }
        {   // explicit graph, two kernel nodes, NO edge
            hipGraph_t g; hipGraphExec_t ge; CK(hipGraphCreate(&g, 0));
            hipGraphNode_t na, nb;
            int baseA = 0, baseB = nA; long long ta = tA, tb = tB;
            void* argsA[] = {&d, &baseA, &ta}; void* argsB[] = {&d, &baseB, &tb};
            hipKernelNodeParams pa = {}; pa.func = (void*)spin_kernel; pa.gridDim = dim3(nA); pa.blockDim = dim3(thr); pa.kernelParams = argsA;
            hipKernelNodeParams pb = pa; pb.gridDim = dim3(nB); pb.kernelParams = argsB;
            CK(hipGraphAddKernelNode(&na, g, nullptr, 0, &pa));
            CK(hipGraphAddKernelNode(&nb, g, nullptr, 0, &pb));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) { CK(hipGraphLaunch(ge, st)); fetch(); }
            report("graph, two kernel nodes, no edge", h, nA, nB);
        }
        {   // two streams
            hipStream_t s2; CK(hipStreamCreate(&s2));
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(spin_kernel, dim3(nA), dim3(thr), 0, st, d, 0, tA);
                hipLaunchKernelGGL(spin_kernel, dim3(nB), dim3(thr), 0, s2, d, nA, tB);
                CK(hipStreamSynchronize(s2)); fetch();
            }
            report("two streams", h, nA, nB);
            CK(hipStreamDestroy(s2));
        }
    }
    return 0;
}
