// LAB kernels for tools/aql_lab.cpp (compiled to a code object: hipcc --offload-arch=gfx950 -O2 --cuda-device-only --no-gpu-bundle-output).  No gridDim / blockDim: nothing may
// depend on hipcc's hidden kernel arguments, the packets are written by hand.
#include <hip/hip_runtime.h>

// out[2*(base+wg)] = start, +1 = end (100 MHz wall clock).  If wait != null: spin (bounded, ~5 ms) until *wait >= wait_val (agent-scope acquire),
// then read data[wg % ndata] (written by the signalling kernel) and put the time of that into seen[base+wg].  If sig != null: every workgroup
// writes data[wg] = stamp, then adds 1 to *sig with agent-scope release.
extern "C" __global__ void lab_kernel(unsigned long long* out, int base, long long ticks, unsigned* wait, unsigned wait_val, unsigned* sig,
                                      unsigned* data, int ndata, unsigned stamp, unsigned long long* seen, unsigned* bad) {
    const int wg = __builtin_amdgcn_workgroup_id_x();
    unsigned long long t0 = wall_clock64();
    if (wait) {
        unsigned v = 0;
        for (int i = 0; i < 2000000; ++i) {
            v = __hip_atomic_load(wait, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= wait_val || (long long)(wall_clock64() - t0) > 500000) break;
            __builtin_amdgcn_s_sleep(2);
        }
        unsigned long long ts = wall_clock64();
        unsigned d = data[(wg + 37) % ndata];
        unsigned long long td = wall_clock64();
        if (threadIdx.x == 0) { seen[2 * (base + wg)] = ts; seen[2 * (base + wg) + 1] = td; if (v < wait_val || d != stamp) atomicAdd(bad, 1u); }
    }
    unsigned long long t1 = wall_clock64();
    for (int i = 0; i < 4000000 && (long long)(t1 - t0) < ticks; ++i) { __builtin_amdgcn_s_sleep(4); t1 = wall_clock64(); }
    if (sig) {
        if (threadIdx.x == 0) data[wg % ndata] = stamp;
        __syncthreads();
        t1 = wall_clock64();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) { out[2 * (base + wg)] = t0; out[2 * (base + wg) + 1] = t1; }
}
