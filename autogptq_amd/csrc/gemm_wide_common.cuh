// gemm_wide_common.cuh -- device helpers shared by the 128 x 128-per-wave prefill kernels (gemm_wide.hip: whole tiles; gemm_wide_sk.hip: stream-K units):
// the exact magic-number dequantisation of four adjacent columns (Deq4), the 32x32x16 matrix-core step (Mma) and the group-constant record (CRaw).
#pragma once
#include <type_traits>

#include "common.cuh"

namespace gptq {
namespace wide {

__device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask, unsigned orv) {
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));      // safe here: all 256 accumulator registers are live
    return r;
}
__device__ __forceinline__ unsigned f16x2_bits(f16x2 v) { return __builtin_bit_cast(unsigned, v); }

template <typename T> struct Mma;
template <> struct Mma<f16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// group constants of a lane's 4 columns: 4 scales (8 bytes) + the qzeros word holding its 4 nibbles
struct CRaw { u32x2 s; unsigned z; };

template <typename T> struct Deq4;
template <> struct Deq4<f16> {
    f16x2 s2[4], c1[4], c2[4];
    __device__ __forceinline__ void setup(const CRaw& c, unsigned zsh, unsigned zmask) {
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const unsigned sw = c.s[col >> 1];
            const unsigned sb = (col & 1) ? (sw >> 16) : (sw & 0xffffu);
            s2[col] = as_f16x2(sb * 0x00010001u);
            const unsigned z = (((c.z >> (zsh + 4 * col)) & 15u) + 1u) & zmask;
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);               // -(1024 + z)
            c2[col] = c1[col] + k960;                                        // -(64 + z), exact
        }
    }
    __device__ __forceinline__ u32x4 frag(unsigned q, int col) const {
#if defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 1)
        return u32x4{q, q ^ 0x11111111u, q ^ 0x22222222u, q ^ f16x2_bits(s2[col])};      // lab: no dequant math (wrong results by construction)
#endif
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1[col];          // k0,k4 : w - z
        const f16x2 h1 = as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];    // k1,k5
        const f16x2 h2 = as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1[col];         // k2,k6
        const f16x2 h3 = as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];   // k3,k7
        return u32x4{f16x2_bits(h0 * s2[col]), f16x2_bits(h1 * s2[col]), f16x2_bits(h2 * s2[col]), f16x2_bits(h3 * s2[col])};
    }
};
template <> struct Deq4<bf16> {            // w - z exactly in packed fp16, times the scale in fp32 (exact product), ONE rounding to bf16: the reference's W
    f16x2 c1[4], c2[4];
    float s[4];
    __device__ __forceinline__ void setup(const CRaw& c, unsigned zsh, unsigned zmask) {
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const unsigned sw = c.s[col >> 1];
            const unsigned short sb = (unsigned short)((col & 1) ? (sw >> 16) : (sw & 0xffffu));
            s[col] = (float)__builtin_bit_cast(bf16, sb);
            const unsigned z = (((c.z >> (zsh + 4 * col)) & 15u) + 1u) & zmask;
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);
            c2[col] = c1[col] + k960;
        }
    }
    static __device__ __forceinline__ unsigned scaled_pair(f16x2 h, float sc) {
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        float lo, hi;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(sc));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(hb), "v"(sc));
        const bf16x2 v = {(bf16)lo, (bf16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ __forceinline__ u32x4 frag(unsigned q, int col) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1[col];
        const f16x2 h1 = as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];
        const f16x2 h2 = as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1[col];
        const f16x2 h3 = as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];
        return u32x4{scaled_pair(h0, s[col]), scaled_pair(h1, s[col]), scaled_pair(h2, s[col]), scaled_pair(h3, s[col])};
    }
};

__device__ __forceinline__ unsigned short t_bits(f16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ unsigned short t_bits(bf16 v) { return __builtin_bit_cast(unsigned short, v); }

}  // namespace wide
}  // namespace gptq
