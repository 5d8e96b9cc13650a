// gemv_shared.cuh -- device helpers shared by the decode kernels of gemv.hip and gemv_tiled.hip (moved here verbatim from gemv.hip).
#pragma once
#include <type_traits>

#include "common.cuh"
#include "launch.h"

namespace gptq {

// kernel arguments of the checkpoint-layout GEMVs (gemv.hip, gemv_generic.hip)
struct GemvParams {
    const unsigned* qweight;
    const unsigned* qzeros;
    const void* scales;
    const int* g_idx;   // per-k groups (PERK mode) or nullptr
    const int* perm;    // x gather for re-sequenced act-order layers, or nullptr
    const void* bias;
    const void* x;
    void* out;
    float* partial;     // [ksplit][M][N] when ksplit > 1
    int M, K, N, group_size, zero_mode;
    int units_total, units_per_split, chunk_units, ksplit;
    int gu_shift;       // log2(group_size / 8) or -1 (matrix-core 4-bit kernel)
    int pair_off;       // SILU_MUL epilogue: column distance between the gate and the up half (N / 2), else 0
};
// gemv_generic.hip: fp32-math GEMV for any bits / dtype / group structure (fp32 layers, raw act-order g_idx, odd group sizes)
hipError_t launch_gemv_generic(const gptq_layer_t& L, const GemvPlan& pl, const GemvParams& p, hipStream_t st);

// Sum over the 64/LN row slots of a wave (lanes l, l+LN, l+2LN, ...): DPP rotates inside a 16-lane row,
// ds_bpermute across rows.  Every lane ends up with the total.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
template <int LN> __device__ __forceinline__ float row_slot_sum(float v) {
    if constexpr (LN <= 4) v = dpp_add<0x124>(v);     // row_ror:4
    if constexpr (LN <= 8) v = dpp_add<0x128>(v);     // row_ror:8
    if constexpr (LN <= 16) v += __shfl_xor(v, 16, 64);
    if constexpr (LN <= 32) v += __shfl_xor(v, 32, 64);
    return v;
}

// 4x4x4 matrix-core step on packed 2-byte operands (u32x2 = 4 values) for both fp16 and bf16
template <typename T> struct Mma4;
template <> struct Mma4<f16> {
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 run(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned short bits_of_int(int d) { return as_u16((f16)(short)d); }
};
template <> struct Mma4<bf16> {
    typedef short v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 run(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned short bits_of_int(int d) { return (unsigned short)(as_u32((float)d) >> 16); }  // exact
};

// 16 bytes per lane, global -> LDS (destination = lds_dst + lane * 16), nontemporal.  Inline asm on purpose: with the
// builtin, hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) behind EVERY LDS-DMA instruction of a burst (it cannot prove that two
// DMA writes into the one __shared__ array do not overlap), which serialises the burst into dependent round trips.  Hidden
// in asm the instruction is not counted by the compiler's own vmcnt bookkeeping -- that only ever makes its waits for
// ordinary loads longer, never shorter (vmcnt retires in order) -- and the kernel waits for the DMA data explicitly.
__device__ __forceinline__ void dma16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct GemvSeg {
    const unsigned* qweight;
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
    void* out;
    int N;         // columns of this layer
    int blk_end;   // cumulative strip count up to and including this layer
    int col0;      // first column of this layer in the concatenated partial slab
    int pad_;
};
struct GemvStreamParams {
    GemvSeg seg[4];
    const void* x;
    unsigned long long* gran;   // [ksplit - 1][M][nsum] exchange granules {fp32 partial sum, tag} of K slices 1 .. ksplit - 1
    unsigned* epochs;           // [strips_total] per-strip launch epoch (header bytes 32768 ..): bumped by the strip's owner slice, never reset
    unsigned* err;              // sticky error word (header tail): a bounded wait gave up
    int nseg, M, K, zero_mode, units_total, units_per_split, ksplit, gu_shift, nsum;
    unsigned max_spins;
};

// Shared tail of the streamed GEMV kernels: row slots (DPP / bpermute), waves (LDS, the kernel's only barrier), then write -- or, with a K split,
// publish / combine through {fp32, tag} granules.
// (stage: optional LDS copy of the finished [MT][CT] outputs of the strip; sg.out may then be null.)
// Second half of the streamed kernels' tail: the per-wave partial sums of the workgroup's CT columns x MT rows are in LDS (red[wave * (MT * CT + 4) + m * CT + c],
// behind a barrier); cross-wave sum, then write -- or, with a K split, publish / combine through {fp32, tag} granules.
template <int CT, int MT, typename T, typename PP = GemvStreamParams, typename SG = GemvSeg>
__device__ __forceinline__ void stream_finish(const PP& p, const SG& sg, int strip, int sidx, int ks, int N, const float* red, T* stage = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    constexpr int E = MT * CT, ES = E + 4;                                    // slab stride padded by 16 B: the W partials of an entry spread over banks
    // K split: slice ks >= 1 PUBLISHES its partial sums as 8-byte {fp32, tag} granules (one write-through store each) and is done; slice 0,
    // the strip's OWNER, polls the granules of the other slices for its entries, takes each the moment its tag is this launch's, adds them in
    // slice order (fixed order: bit-reproducible) and writes the result -- ONE memory hop after the last slice has published, where the
    // ticket scheme this replaces (publish, drain, draw a ticket, last arriver reads everything back) was three (~3 us).
    // tag = the strip's epoch word + 1 in a NaN pattern; the owner bumps the word when all its waves are through (by then every producer
    // wave has read it), so the next launch on this workspace -- any layer -- uses a tag that no stale granule carries.
    unsigned tag = 0;
    if (p.ksplit > 1) {
        unsigned ep;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ep) : "s"(p.epochs + sidx) : "memory");
        tag = 0x7FE00000u | ((ep + 1u) & 0x1FFFFFu);
    }
    const size_t slab = (size_t)p.M * p.nsum;
    bool gave_up = false;
    auto emit = [&](int e, float t) {
        const int m = e / CT, c = e % CT;
        const int n = strip * CT + c;
        if (n >= N || m >= p.M) return;
        if (p.ksplit > 1) {
            const size_t at = (size_t)m * p.nsum + sg.col0 + n;
            if (ks != 0) {
                const unsigned long long g8 = (unsigned long long)as_u32(t) | ((unsigned long long)tag << 32);
                __hip_atomic_store(p.gran + (size_t)(ks - 1) * slab + at, g8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            constexpr int KMAX = 8;                                           // planner: ksplit <= 8
            unsigned long long v[KMAX - 1];
            unsigned pending = (1u << (p.ksplit - 1)) - 1u;
            for (unsigned spins = 0; pending; ++spins) {
#pragma unroll
                for (int k = 0; k < KMAX - 1; ++k)
                    if (pending & (1u << k)) v[k] = __hip_atomic_load(p.gran + (size_t)k * slab + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < KMAX - 1; ++k)
                    if ((pending & (1u << k)) && (unsigned)(v[k] >> 32) == tag) pending &= ~(1u << k);
                if (pending && spins > p.max_spins) { gave_up = true; break; }
                if (pending) __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int k = 0; k < KMAX - 1; ++k)
                if (k < p.ksplit - 1) t += as_f32((unsigned)(v[k] & 0xffffffffu));
            // consumed granules are cleared: between launches the exchange area holds NO valid tag, so a tag that is valid now was written by
            // this launch -- per-strip epochs alone would let a strip whose epoch lags (it is used by fewer layers of the model) accept what
            // another layer published at the same address under the same number
#pragma unroll
            for (int k = 0; k < KMAX - 1; ++k)
                if (k < p.ksplit - 1) __hip_atomic_store(p.gran + (size_t)k * slab + at, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (sg.bias) t += DType<T>::to_f32(((const T*)sg.bias)[n]);
        const T o = DType<T>::from_f32(t);
        if (sg.out) ((T*)sg.out)[(size_t)m * N + n] = o;
        if (stage) stage[m * CT + c] = o;                                      // tensor-parallel epilogue (gemv_tiled.hip): the strip's rows, for the peer stores
    };
    if ((W & (W - 1)) == 0) {
        // every wave takes 64 / W entries per round; its lanes are (entry, partial w) pairs: one LDS read each, then a fixed
        // xor tree over the W lanes of an entry (deterministic order) -- instead of one thread walking W slabs per entry
        const int lw = __builtin_ctz((unsigned)W);
        const int w_of_lane = lane & (W - 1), e_of_lane = lane >> lw;
        for (int e0 = 0; e0 < E; e0 += 64) {
            const int e = e0 + wave * (64 >> lw) + e_of_lane;
            float t = (e < E) ? red[w_of_lane * ES + e] : 0.f;
            for (int off = 1; off < W; off <<= 1) t += __shfl_xor(t, off, 64);
            if (w_of_lane == 0 && e < E) emit(e, t);
        }
    } else {
        for (int e = tid; e < E; e += blockDim.x) {
            float t = 0.f;
            for (int w = 0; w < W; ++w) t += red[w * ES + e];
            emit(e, t);
        }
    }
    if (p.ksplit > 1 && ks == 0) {
        if (gave_up) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                                      // every wave of the owner has its granules: every producer wave has read the epoch
        if (tid == 0) __hip_atomic_store(p.epochs + sidx, (tag & 0x1FFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int LN, int MT, typename T>
__device__ __forceinline__ void stream_epilogue(f32x4 (&acc)[MT], const GemvStreamParams& p, const GemvSeg& sg, int strip, int sidx, int ks, int N,
                                                float* red) {
    constexpr int CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    // ---- row slots (DPP / bpermute), waves (LDS, the only barrier), then write or publish ------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = row_slot_sum<LN>(acc[m][c]);
    constexpr int E = MT * CT, ES = E + 4;                                    // slab stride padded by 16 B: the W partials of an entry spread over banks
    if (lane < LN) {
#pragma unroll
        for (int m = 0; m < MT; ++m) *(f32x4*)(red + wave * ES + m * CT + lane * 4) = acc[m];
    }
    __syncthreads();
    stream_finish<CT, MT, T>(p, sg, strip, sidx, ks, N, red);
}


}  // namespace gptq
