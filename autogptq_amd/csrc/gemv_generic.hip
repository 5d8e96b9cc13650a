// gemv_generic.hip -- the fp32-math GEMV on the checkpoint layout: any bits / dtype / group structure.  Default plan for fp32 layers, raw (non-uniform)
// act-order g_idx without a re-sequenced copy, group sizes that are not whole packing units, and 4-bit fp16 layers whose groups are not a power of two.
// Replaces (reference): VecQuant{2,3,4,8}MatMulKernel in autogptq_extension/cuda_256/autogptq_cuda_kernel_256.cu:281-1437 (per-k g_idx lookups, fp32 FMA).
// A workgroup owns a column strip of 4 LN columns over a K range, x staged in LDS per K chunk; (split out of gemv.hip in round 6: its own translation unit,
// only the instantiations plan_gemv can ask for).
#include <type_traits>
#include <utility>

#include "common.cuh"
#include "launch.h"
#include "gemv_shared.cuh"

namespace gptq {

// ---- shared epilogue: reduce row slots (shuffles), waves (LDS), then write ------------------
template <typename T, int LN, int MT>
__device__ __forceinline__ void reduce_and_store(float (&acc)[MT][4], float* red, const GemvParams& p,
                                                 int strip, int m0) {
    constexpr int CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[m][c];
#pragma unroll
            for (int off = LN; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            acc[m][c] = v;
        }
    __syncthreads();  // everyone is done reading the x chunk that aliases `red`
    if (lane < LN) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(wave * MT + m) * CT + lane * 4 + c] = acc[m][c];
    }
    __syncthreads();
    for (int i = tid; i < MT * CT; i += blockDim.x) {
        const int m = i / CT, c = i % CT;
        const int n = strip * CT + c, row = m0 + m;
        if (n >= p.N || row >= p.M) continue;
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += red[(w * MT + m) * CT + c];
        if (p.ksplit > 1) {
            p.partial[((size_t)blockIdx.y * p.M + row) * p.N + n] = s;
        } else {
            if (p.bias) s += DType<T>::to_f32(((const T*)p.bias)[n]);
            ((T*)p.out)[(size_t)row * p.N + n] = DType<T>::from_f32(s);
        }
    }
}

// ---- generic kernel: any bits / dtype / group structure, fp32 math -------------------------
// PERK = false: one (scale, zero) per packed unit and column (sequential groups, unit inside a
//               group);  PERK = true: group looked up per k (raw act-order g_idx, odd group sizes).
template <int BITS, typename T, int LN, int MT, bool PERK>
__global__ void __launch_bounds__(1024) gemv_generic_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals, WR = 64 / LN, CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = strip * CT + cl * 4;
    const bool col_ok = n0 < p.N;
    const int m0 = blockIdx.z * MT;
    const int ub = blockIdx.y * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    const int xstride = p.chunk_units * KPU;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ scales = (const T*)p.scales;
    const int zrow_words = p.N / 32 * BITS;

    float acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = 0.f;

    for (int cb = ub; cb < ue; cb += p.chunk_units) {
        const int ce = min(cb + p.chunk_units, ue);
        const int kc = (ce - cb) * KPU, kbase = cb * KPU;
        if (cb != ub) __syncthreads();
        for (int i = tid; i < MT * kc; i += blockDim.x) {
            const int m = i / kc, kk = i - m * kc;
            const int k = kbase + kk;
            const int src = p.perm ? p.perm[k] : k;
            xs[m * xstride + kk] = (m0 + m < p.M) ? DType<T>::to_f32(x[(size_t)(m0 + m) * p.K + src]) : 0.f;
        }
        __syncthreads();
        for (int it = 0;; ++it) {
            const int ubase = cb + (it * W + wave) * WR;
            if (ubase >= ce) break;
            const int u = ubase + rs;
            if (u >= ce || !col_ok) continue;
            u32x4 q[UW];
#pragma unroll
            for (int w = 0; w < UW; ++w)
                q[w] = *(const u32x4*)(p.qweight + (size_t)(u * UW + w) * p.N + n0);
            const int k0 = u * KPU;
            const float* xk = xs + (k0 - kbase);
            if constexpr (!PERK) {
                const int g = k0 / p.group_size;
                float s[4];
                int z[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) s[c] = DType<T>::to_f32(scales[(size_t)g * p.N + n0 + c]);
                zero_points4(p.qzeros + (size_t)g * zrow_words, n0, BITS, p.zero_mode, z);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned w[UW];
#pragma unroll
                    for (int i = 0; i < UW; ++i) w[i] = q[i][c];
                    float d[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) d[m] = 0.f;
                    // w - z in integers (exact), like the reference's (weight - zeros): a layer whose fields equal their
                    // zero-point gives exactly 0, with no cancellation between sum(x*w) and z*sum(x)
                    [&]<int... V>(std::integer_sequence<int, V...>) {
                        (([&] {
                             const float wf = (float)((int)unit_field<BITS, V>(w) - z[c]);
#pragma unroll
                             for (int m = 0; m < MT; ++m) d[m] = fmaf(xk[m * xstride + V], wf, d[m]);
                         }()),
                         ...);
                    }(std::make_integer_sequence<int, KPU>{});
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][c] = fmaf(s[c], d[m], acc[m][c]);
                }
            } else {
                // per-k groups: a ROLLED loop over the unit's values (runtime field position).  Unrolled over all 32 values of a 3-bit unit the compiler
                // serialised the g_idx -> (scales, zeros) chains AND parked them in scratch (136-390 spilled registers under the 128-register cap of a
                // 16-wave workgroup); four values per trip keep four chains in flight in ~60 registers.  Same fma order per accumulator as before.
#pragma unroll 4
                for (int v = 0; v < KPU; ++v) {
                    const int k = k0 + v;
                    const int g = p.g_idx ? p.g_idx[k] : k / p.group_size;
                    int z[4];
                    zero_points4(p.qzeros + (size_t)g * zrow_words, n0, BITS, p.zero_mode, z);
                    const unsigned bit = (unsigned)BITS * (unsigned)v, wi = bit >> 5, sh = bit & 31;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        unsigned f;
                        if constexpr (UW == 1) {
                            f = (q[0][c] >> sh) & Pack<BITS>::maxq;
                        } else {                // 3 bits: the unit is a 96-bit little-endian stream, values 10 and 21 straddle a word
                            const unsigned lo = wi == 0 ? q[0][c] : (wi == 1 ? q[1][c] : q[2][c]);
                            const unsigned hi = wi == 0 ? q[1][c] : (wi == 1 ? q[2][c] : 0u);
                            f = (unsigned)(((((unsigned long long)hi) << 32) | lo) >> sh) & Pack<BITS>::maxq;
                        }
                        const float s = DType<T>::to_f32(scales[(size_t)g * p.N + n0 + c]);
                        const float dq = s * (float)((int)f - z[c]);
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m][c] = fmaf(xk[m * xstride + v], dq, acc[m][c]);
                    }
                }
            }
        }
    }
    reduce_and_store<T, LN, MT>(acc, (float*)smem, p, strip, m0);
}

// Only what plan_gemv asks for: strips of 16 or 64 columns (LN = 4 / 16; fp32 layers -- this kernel's main users -- also 32: 4096 -> 11008 M = 1 / 4
// 21.3 / 35.0 us with 16-column strips, 20.1 / 32.4 with 32, tools/fallback_ab.py); 3-bit units hold 32 values -- one row of x per pass with per-k groups, two without.
template <int BITS, typename T, int LN, int MT>
static hipError_t launch_generic_ln(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    dim3 grid(pl.strips, pl.ksplit, pl.mtiles), block(pl.waves * 64);
    if (pl.perk) {
        if constexpr (BITS == 3 && MT > 1) return hipErrorInvalidValue;
        else hipLaunchKernelGGL((gemv_generic_kernel<BITS, T, LN, MT, true>), grid, block, pl.lds_bytes, st, p);
    } else {
        hipLaunchKernelGGL((gemv_generic_kernel<BITS, T, LN, MT, false>), grid, block, pl.lds_bytes, st, p);
    }
    return hipGetLastError();
}

template <int BITS, typename T, int MT>
static hipError_t launch_generic_mt(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (pl.ln) {
        case 4: return launch_generic_ln<BITS, T, 4, MT>(pl, p, st);
        case 8: if constexpr (std::is_same_v<T, float>) return launch_generic_ln<BITS, T, 8, MT>(pl, p, st); else return hipErrorInvalidValue;
        case 16: return launch_generic_ln<BITS, T, 16, MT>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

template <int BITS, typename T>
static hipError_t launch_generic_bits(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (pl.mt) {
        case 1: return launch_generic_mt<BITS, T, 1>(pl, p, st);
        case 2: return launch_generic_mt<BITS, T, 2>(pl, p, st);
        case 4: if constexpr (BITS != 3) return launch_generic_mt<BITS, T, 4>(pl, p, st); else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

template <typename T>
static hipError_t launch_generic(const gptq_layer_t& L, const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (L.bits) {
        case 2: return launch_generic_bits<2, T>(pl, p, st);
        case 3: return launch_generic_bits<3, T>(pl, p, st);
        case 4: return launch_generic_bits<4, T>(pl, p, st);
        case 8: return launch_generic_bits<8, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemv_generic(const gptq_layer_t& L, const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (L.dtype) {
        case GPTQ_F16: return launch_generic<f16>(L, pl, p, st);
        case GPTQ_BF16: return launch_generic<bf16>(L, pl, p, st);
        case GPTQ_F32: return launch_generic<float>(L, pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gptq
