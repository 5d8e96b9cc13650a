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

// ---- 3-bit and 8-bit forms (round 5: the stream-K kernel on BASELINE config 5) -----------------------------------------------------------------------
// Constants come from the decode copy's records (qconst_tiled: 16 scales + 16 zero-points AS USED per strip and group): 4 scales (8 bytes) and the
// zero-points of the lane's 4 columns -- one byte each at 3 bits (z <= 8), 16 bits each at 8 bits (z <= 256).
struct CRaw8 { u32x2 s; u32x2 z; };
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ unsigned bf16_scaled_pair(f16x2 h, float sc) {      // (w - z) exact in fp16, times the scale in fp32, ONE rounding to bf16
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    float lo, hi;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(sc));
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(hb), "v"(sc));
    const bf16x2 v = {(bf16)lo, (bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
template <typename T> struct Scale4;           // the 4 columns' scales and the last step of every pair: (w - z) * scale, rounded once to T
template <> struct Scale4<f16> {
    f16x2 s2[4];
    __device__ __forceinline__ void setup(u32x2 s) {
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const unsigned sw = s[col >> 1];
            s2[col] = as_f16x2(((col & 1) ? (sw >> 16) : (sw & 0xffffu)) * 0x00010001u);
        }
    }
    __device__ __forceinline__ unsigned mul(f16x2 h, int col) const { return f16x2_bits(h * s2[col]); }
};
template <> struct Scale4<bf16> {
    float s[4];
    __device__ __forceinline__ void setup(u32x2 sv) {
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const unsigned sw = sv[col >> 1];
            s[col] = (float)__builtin_bit_cast(bf16, (unsigned short)((col & 1) ? (sw >> 16) : (sw & 0xffffu)));
        }
    }
    __device__ __forceinline__ unsigned mul(f16x2 h, int col) const { return bf16_scaled_pair(h, s[col]); }
};

// 3 bits (decode copy, utils.hip prepack_decode_weights_kernel<3>): word j of a lane's three holds pair 5 j + i = (k 2p, k 2p + 1) at bit 3 i of its low / high
// half, i = 0..4; bit 15 / 31 of word j = bit j of k30 / k31.  A field inside the fp16 mantissa (bits 0..9) is read in place: OR-ing the exponent of 1024
// gives 1024 + 2^(3i) w, and one fma with 2^(-3i) and -(2^(10-3i) + z) leaves w - z exactly; fields i = 3, 4 are read from q >> 6 as i = 1, 2.
template <typename T> struct Deq3 {
    Scale4<T> sc;
    f16x2 c0[4], c1[4], c2[4];
    __device__ __forceinline__ void setup(const CRaw& c) {
        sc.setup(c.s);
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const unsigned z = (c.z >> (8 * col)) & 0xffu;
            c0[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);                 // -(1024 + z)
            c1[col] = as_f16x2(z * 0x00080008u + 0xD800D800u);                 // -(128 + z): the ulp at 128 is 1/8
            c2[col] = as_f16x2(z * 0x00400040u + 0xCC00CC00u);                 // -(16 + z): the ulp at 16 is 1/64
        }
    }
    __device__ __forceinline__ f16x2 p0(unsigned q, int col) const { return as_f16x2(and_or(q, 0x00070007u, 0x64006400u)) + c0[col]; }
    __device__ __forceinline__ f16x2 p1(unsigned q, int col) const {
        const f16x2 r = {(f16)0.125f, (f16)0.125f};
        return as_f16x2(and_or(q, 0x00380038u, 0x64006400u)) * r + c1[col];
    }
    __device__ __forceinline__ f16x2 p2(unsigned q, int col) const {
        const f16x2 r = {(f16)0.015625f, (f16)0.015625f};
        return as_f16x2(and_or(q, 0x01c001c0u, 0x64006400u)) * r + c2[col];
    }
    // the 8 values k = 8 ks .. 8 ks + 7 of the lane's 32 (pairs 4 ks .. 4 ks + 3); ks is a constant after unrolling
    __device__ __forceinline__ u32x4 frag(const u32x3& w, int ks, int col) const {
        f16x2 h[4];
        if (ks == 0) {
            h[0] = p0(w[0], col); h[1] = p1(w[0], col); h[2] = p2(w[0], col); h[3] = p1(w[0] >> 6, col);
        } else if (ks == 1) {
            h[0] = p2(w[0] >> 6, col); h[1] = p0(w[1], col); h[2] = p1(w[1], col); h[3] = p2(w[1], col);
        } else if (ks == 2) {
            const unsigned q6 = w[1] >> 6;
            h[0] = p1(q6, col); h[1] = p2(q6, col); h[2] = p0(w[2], col); h[3] = p1(w[2], col);
        } else {
            const unsigned q6 = w[2] >> 6;
            unsigned t = (w[0] >> 15) & 0x00010001u;
            t = and_or(w[1] >> 14, 0x00020002u, t);
            t = and_or(w[2] >> 13, 0x00040004u, t);
            h[0] = p2(w[2], col); h[1] = p1(q6, col); h[2] = p2(q6, col); h[3] = p0(t, col);
        }
        return u32x4{sc.mul(h[0], col), sc.mul(h[1], col), sc.mul(h[2], col), sc.mul(h[3], col)};
    }
};

// 8 bits: stored byte p of word w = k 4 w + {0, 2, 1, 3}[p]: (q & 0x00ff00ff) = (k0, k1), the same on q >> 8 = (k2, k3); 1024 + w - (1024 + z) is exact
template <typename T> struct Deq8 {
    Scale4<T> sc;
    f16x2 c1[4];
    __device__ __forceinline__ void setup(const CRaw8& c) {
        sc.setup(c.s);
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const unsigned zw = c.z[col >> 1];
            const unsigned z = (col & 1) ? (zw >> 16) : (zw & 0xffffu);
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);
        }
    }
    // the 8 values of one MFMA step: two adjacent words of the lane's eight
    __device__ __forceinline__ u32x4 frag(unsigned q0, unsigned q1, int col) const {
        const f16x2 h0 = as_f16x2(and_or(q0, 0x00ff00ffu, 0x64006400u)) + c1[col];
        const f16x2 h1 = as_f16x2(and_or(q0 >> 8, 0x00ff00ffu, 0x64006400u)) + c1[col];
        const f16x2 h2 = as_f16x2(and_or(q1, 0x00ff00ffu, 0x64006400u)) + c1[col];
        const f16x2 h3 = as_f16x2(and_or(q1 >> 8, 0x00ff00ffu, 0x64006400u)) + c1[col];
        return u32x4{sc.mul(h0, col), sc.mul(h1, col), sc.mul(h2, col), sc.mul(h3, col)};
    }
};

__device__ __forceinline__ unsigned short t_bits(f16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ unsigned short t_bits(bf16 v) { return __builtin_bit_cast(unsigned short, v); }

}  // namespace wide
}  // namespace gptq
