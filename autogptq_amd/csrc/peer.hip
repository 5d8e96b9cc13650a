// peer.hip -- direct peer-store all-gather for out_features-parallel layers ("each rank writes its [M, N/T]
// slice straight into the 7 peers' output buffers over xGMI ... beats a ring").
//
// The reference has no multi-GPU execution of a layer (tests/test_q4.py:1224-1226 `test_multigpu` is a TODO); the column split
// itself is the fact its fused-QKV caller relies on (fused_llama_attn.py:171-186).  xGMI is point-to-point (7 links per GPU,
// one per peer), so the all-gather of a decode-sized slice (N/T columns x 2 bytes = 1-7 KB per row) is T - 1 independent
// stores, one per link, with no ring step: every rank stores its slice at its column offset of EVERY rank's exchange buffer
// (its own included) and then raises its arrival flag in every rank's flag array; a rank's gathered rows are complete when
// all T of its flags carry the call's epoch.
//
// Two launches per gather on the caller's stream; no host-side state changes per call, so a captured hipGraph replays it:
//   peer_scatter_kernel    16-byte payload stores, lanes contiguous along a row of the slice; every block then fences at system
//                          scope and takes a ticket, the last block publishes epoch e + 1 into flags[r][rank] of every rank r
//                          (system-scope release) and resets the ticket.
//   (round 4) gptq_forward_scatter makes the payload stores the EPILOGUE of the rank's decode kernel (gemv_tiled.hip: write-through stores, no fence, no
//                          ticket -- both measured at 8 .. 240 us per launch when every strip does one); the rank's flag is then raised by the first
//                          block of its collect launch, which starts behind that kernel in stream order (peer_publish_kernel does only that).
//   peer_collect_kernel    every block: lanes 0..T-1 poll flags[rank][r] (relaxed, system scope) until they carry the epoch --
//                          a BOUNDED spin (max_spins polls, then state[3] is raised and the kernel carries on instead of
//                          hanging the queue) -- then a system-scope acquire, then the block copies its share of the gathered
//                          rows from the exchange buffer into the caller's `out`; the last block (ticket) advances the epoch.
// The epoch lives in device memory (state[0] = gathers completed) and both kernels read it, which is what lets a graph replay.
// Two exchange buffers alternate by epoch parity: a rank can be at most one epoch ahead of the slowest peer (its collect of epoch
// e needs every peer's scatter of e, which that peer enqueues behind its own collect of e - 1), so the buffer of epoch e + 1 is
// never one a peer is still copying out of.  The copy into `out` is what buys the fixed output address a captured consumer
// needs; it is M x N x 2 bytes of local traffic (8-16 KB for a decode row).
//
// Memory-model note: flags and exchange buffers are written by other agents while this agent's kernels run, i.e. they must be
// fine-grained (or uncached) allocations for the system-scope fences to mean anything across GPUs.  On one device (peers =
// other buffers, or other processes' buffers mapped through IPC) any allocation works; that is the only configuration this
// code could be exercised in (1-GPU boxes), and it is off by default (ColumnParallelQuantLinear(..., exchange="peer_store")).
#include <algorithm>

#include "common.cuh"
#include "launch.h"

namespace gptq {

struct PeerArgs {
    char* xbuf[2][GPTQ_PEER_MAX];
    unsigned* flags[GPTQ_PEER_MAX];
    unsigned* state;     // [0] epochs completed, [1] scatter ticket, [2] collect ticket, [3] timeout raised
    int world, rank;
};

// grid = (gx, world): block column y serves peer y
__global__ void __launch_bounds__(256) peer_scatter_kernel(PeerArgs a, const u32x4* __restrict__ y, int M, int chunks_per_row,
                                                           size_t row_stride_bytes, size_t col_off_bytes) {
    const unsigned e = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;     // this gather's epoch
    const size_t total = (size_t)M * chunks_per_row;
    char* const dst = a.xbuf[e & 1u][blockIdx.y] + col_off_bytes;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / chunks_per_row, c = i - m * chunks_per_row;
        *(u32x4*)(dst + m * row_stride_bytes + c * 16) = y[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: this thread's payload stores are complete (peers included)
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        const unsigned nblk = gridDim.x * gridDim.y;
        last = (__hip_atomic_fetch_add(a.state + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1u);
    }
    __syncthreads();
    if (last) {
        if ((int)threadIdx.x < a.world)
            __hip_atomic_store(a.flags[threadIdx.x] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256) peer_collect_kernel(PeerArgs a, u32x4* __restrict__ out, size_t chunks, unsigned max_spins) {
    const unsigned e = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // The rank's arrival flag, (re-)published by the first block: everything this rank enqueued in front of this launch has completed (stream order), its
    // payload stores included -- this is what raises the flag when the scatter was the epilogue of the rank's decode kernel (gptq_forward_scatter, which
    // has no ticket of its own); behind peer_scatter_kernel it stores the value that kernel already stored.
    if (blockIdx.x == 0 && (int)threadIdx.x < a.world)
        __hip_atomic_store(a.flags[threadIdx.x] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < a.world) {
        const unsigned* f = a.flags[a.rank] + threadIdx.x;
        unsigned spins = 0;
        // arrived <=> flag - epoch >= 0 in wrap-around arithmetic (a peer may already be one epoch ahead)
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            if (++spins >= max_spins) {
                __hip_atomic_store(a.state + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");        // system scope: the copy below must not be served from lines older than the flags
    __syncthreads();
    const u32x4* __restrict__ src = (const u32x4*)a.xbuf[e & 1u][a.rank];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) out[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(a.state + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(a.state + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.state, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);       // every block has read the old epoch
        }
    }
}

// The rank's arrival flag on its own: for callers that run several ranks on ONE stream (the simulated-rank tests) or enqueue unrelated work between a
// gptq_forward_scatter and its collect.  Everything the rank enqueued in front has completed (stream order).
__global__ void __launch_bounds__(64) peer_publish_kernel(PeerArgs a) {
    const unsigned e = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if ((int)threadIdx.x < a.world) __hip_atomic_store(a.flags[threadIdx.x] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static PeerArgs peer_args(const gptq_peer_group_t& pg) {
    PeerArgs a{};
    for (int r = 0; r < pg.world; ++r) {
        a.xbuf[0][r] = (char*)pg.xbuf[0][r];
        a.xbuf[1][r] = (char*)pg.xbuf[1][r];
        a.flags[r] = pg.flags[r];
    }
    a.state = pg.state;
    a.world = pg.world;
    a.rank = pg.rank;
    return a;
}

hipError_t launch_peer_scatter(const gptq_peer_group_t& pg, const void* y_local, int M, int n_local, int dtype, hipStream_t st) {
    const PeerArgs a = peer_args(pg);
    const int es = dtype_size(dtype);
    const int cpr = n_local * es / 16;
    const size_t total = (size_t)M * cpr;
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 1024));
    hipLaunchKernelGGL(peer_scatter_kernel, dim3(gx, pg.world), dim3(256), 0, st, a, (const u32x4*)y_local, M, cpr,
                       (size_t)pg.N * es, (size_t)pg.rank * n_local * es);
    return hipGetLastError();
}

hipError_t launch_peer_publish(const gptq_peer_group_t& pg, hipStream_t st) {
    hipLaunchKernelGGL(peer_publish_kernel, dim3(1), dim3(64), 0, st, peer_args(pg));
    return hipGetLastError();
}

hipError_t launch_peer_collect(const gptq_peer_group_t& pg, void* out, int M, int dtype, unsigned max_spins, hipStream_t st) {
    const PeerArgs a = peer_args(pg);
    const size_t chunks = (size_t)M * pg.N * dtype_size(dtype) / 16;
    // every block polls: keep the pollers few (each is T system-scope loads per round) -- one block per 64 KB of output, at most 256
    const unsigned gc = (unsigned)std::max<size_t>(1, std::min<size_t>((chunks + 4095) / 4096, 256));
    hipLaunchKernelGGL(peer_collect_kernel, dim3(gc), dim3(256), 0, st, a, (u32x4*)out, chunks, max_spins);
    return hipGetLastError();
}

}  // namespace gptq
