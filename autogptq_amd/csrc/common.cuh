// common.cuh -- shared device helpers for the gfx950 GPTQ kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gptq_mi355x.h"

namespace gptq {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// Bit casts take their operand BY VALUE on purpose: hipcc (ROCm 7.2 / clang 22) miscompiles
// __builtin_bit_cast(T, vec[i]) on an ext_vector element -- every i reads element 0.
__device__ __forceinline__ f16x2 as_f16x2(unsigned v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ float as_f32(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned as_u32(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ f16 as_f16(unsigned short v) { return __builtin_bit_cast(f16, v); }
__device__ __forceinline__ unsigned short as_u16(f16 v) { return __builtin_bit_cast(unsigned short, v); }

// ---- dtype traits ---------------------------------------------------------------------------
template <typename T> struct DType;
template <> struct DType<f16> {
    static constexpr int id = GPTQ_F16;
    static __device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
    static __device__ __forceinline__ f16 from_f32(float v) { return (f16)v; }  // RNE
};
template <> struct DType<bf16> {
    static constexpr int id = GPTQ_BF16;
    static __device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
    static __device__ __forceinline__ bf16 from_f32(float v) { return (bf16)v; }  // RNE
};
template <> struct DType<float> {
    static constexpr int id = GPTQ_F32;
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};

__host__ __device__ constexpr int dtype_size(int dt) { return dt == GPTQ_F32 ? 4 : 2; }

// ---- packed-field geometry ------------------------------------------------------------------
// A "unit" is the smallest run of whole 32-bit words holding whole values:
//   bits 2/4/8 : 1 word  = 16/8/4 values;   bits 3 : 3 words = 32 values (qlinear_cuda.py:144-162)
template <int BITS> struct Pack {
    static constexpr int words = (BITS == 3) ? 3 : 1;        // words per unit
    static constexpr int vals = (BITS == 3) ? 32 : 32 / BITS; // values per unit
    static constexpr unsigned maxq = (1u << BITS) - 1u;
};
__host__ __device__ constexpr int unit_words(int bits) { return bits == 3 ? 3 : 1; }
__host__ __device__ constexpr int unit_vals(int bits) { return bits == 3 ? 32 : 32 / bits; }

// Field v (compile-time after unrolling) of a unit held in registers, little-endian bit stream.
template <int BITS, int V>
__device__ __forceinline__ unsigned unit_field(const unsigned (&w)[Pack<BITS>::words]) {
    constexpr int bit = BITS * V;
    constexpr int wi = bit >> 5;
    constexpr int sh = bit & 31;
    if constexpr (sh + BITS <= 32) {
        return (w[wi] >> sh) & Pack<BITS>::maxq;
    } else {  // 3-bit straddlers (values 10 and 21)
        return ((w[wi] >> sh) | (w[wi + 1] << (32 - sh))) & Pack<BITS>::maxq;
    }
}

// Field at runtime position v of a bit stream in memory (words `stride` apart).
__device__ __forceinline__ unsigned stream_field(const unsigned* __restrict__ words, size_t stride,
                                                 unsigned v, int bits) {
    const unsigned bit = (unsigned)bits * v;
    const unsigned wi = bit >> 5, sh = bit & 31;
    unsigned long long lo = words[(size_t)wi * stride];
    unsigned long long hi = (sh + (unsigned)bits > 32u) ? words[(size_t)(wi + 1) * stride] : 0ull;
    return (unsigned)(((lo | (hi << 32)) >> sh) & ((1u << bits) - 1u));
}

// Zero-points of 4 adjacent columns n0..n0+3 (n0 % 4 == 0) of one qzeros row, as used in dequant.
__device__ __forceinline__ void zero_points4(const unsigned* __restrict__ zrow, int n0, int bits,
                                             int zero_mode, int (&z)[4]) {
    const unsigned bit = (unsigned)bits * (unsigned)n0;
    const unsigned wi = bit >> 5, sh = bit & 31;
    unsigned long long lo = zrow[wi];
    unsigned long long hi = (sh + 4u * (unsigned)bits > 32u) ? zrow[wi + 1] : 0ull;
    const unsigned long long v = (lo | (hi << 32)) >> sh;
    const unsigned maxq = (1u << bits) - 1u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int f = (int)((v >> (bits * c)) & maxq) + 1;
        z[c] = (zero_mode == GPTQ_ZERO_WRAP) ? (f & (int)maxq) : f;
    }
}

// ---- LDS DMA from inline asm ------------------------------------------------------------------------------------------
// global_load_lds_dwordx4: 16 bytes per lane, global -> LDS at (lds_dst + lane * 16), no VGPR destination.  Issued from asm on
// purpose: with __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) protects every later LDS access that might alias the DMA's
// destination with s_waitcnt vmcnt(0) -- behind EACH DMA of a burst, and in front of the first ds_read of a double-buffered
// K-step, i.e. right after the next step's loads were issued (the whole memory latency lands on every step).  Hidden in asm the
// instruction is not part of the compiler's vmcnt bookkeeping, which only ever makes its waits for ordinary loads longer
// (vmcnt retires in order), never shorter; the kernel waits for the DMA data with a hand-counted s_waitcnt.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst_) {           // default cache policy
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);   // wave-uniform by construction; an SGPR for the compiler too
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// XCD-aware, bijective remap of a linear workgroup id: consecutive logical ids land on the same
// XCD (observed placement: hardware block b runs on XCD b % 8), so neighbouring column strips
// share one L2.  Performance only -- correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace gptq
