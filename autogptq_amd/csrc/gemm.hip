// gemm.hip -- MFMA prefill path: out[M,N] = x[M,K] @ dequant(qweight) for M > 8 (fp16 / bf16).
//
// Replaces (reference, AutoGPTQ v0.8.0.dev0): Marlin<...> in autogptq_extension/marlin/marlin_cuda_kernel.cu:216-727,
// the dequant + cublasHgemm fallbacks of exllama (exllama/cuda_func/q4_matmul.cu:225-260) and exllamav2
// (exllamav2/cuda/q_gemm.cu:104-181), and the torch.matmul branch of the Python classes
// (qlinear_cuda.py:253-317).  Nothing is derived from that code.  The design starts from one property of the
// GPTQ checkpoint layout on CDNA4:
//
//   v_mfma_f32_32x32x16_{f16,bf16} wants, in lane l, the 8 consecutive k values  k0 + 8*(l>>5) .. +7  of column
//   n = l & 31 of B -- and one 4-bit qweight word IS 8 consecutive k of one column.  So the packed weights
//   never go through LDS: each lane loads the words of its own columns straight from the unmodified
//   [K/8, N] tensor (row-contiguous => coalesced), dequantises them in registers and feeds them to the
//   matrix core as the B fragment.  A lane owns 2 adjacent columns (one 8-byte load per packed row), a
//   wave a 64-column strip, a workgroup 4 waves side by side = 256 columns x BM rows; every dequantised
//   word is reused by the MT = BM/32 row tiles of the wave.
//
//   * x (the A operand) is staged through LDS in BK-wide K-steps, double buffered, one barrier per K-step,
//     global loads for step t+1 issued before the MFMAs of step t.  Rows are padded by 16 B, which makes
//     the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots (conflict free).
//   * 4-bit fp16 dequant is the exact magic-number form: (q & 0x000f000f) | 0x64006400 is the half2
//     (1024+w_k, 1024+w_{k+4}); adding -(1024+z) is exact and ONE multiply by the scale rounds exactly like the
//     reference's  scales * (weight - zeros)  (qlinear_cuda_old.py:348).  The words therefore come out in
//     slot order k0,k4,k1,k5,k2,k6,k3,k7 -- x is written to LDS in the same slot order (4 v_perm per 16 B), which
//     is all the MFMA needs (A and B only have to agree on which k sits in which slot).
//     3- and 8-bit fp16 use the same packed form (Deq<3, f16>, Deq<8, f16>: same slot order); every other (bits, dtype) uses fp32 math
//     per field: T(float(s) * float(w - z)) -- also exactly the reference's W.
//   * group_size % BK == 0, so one (scale, zero) pair per column per K-step, fetched one step ahead.
//   * act-order: weights come from the group-sorted side copy (qweight_seq) and x is permuted once per call
//     into the workspace by a small LDS-staged gather kernel (the column_remap of exllama, column_remap.cu:9-63).
//   * fp32 accumulation in the MFMA; optional split-K (only when M*N is too small to fill 256 CUs) writes
//     fp32 partial slabs that a second pass sums in fixed order: bit-reproducible, no atomics.
#include <type_traits>

#include "common.cuh"
#include "launch.h"
#include "../../include/gptq_mi355x_lab.h"

namespace gptq {

struct GemmParams {
    const unsigned* qweight;
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
    const void* x;
    void* out;
    float* partial;   // [ksplit][M][N] fp32 when ksplit > 1
    int M, K, N, group_size, zero_mode;
    int nbm, nbn, ksplit, ksteps_total, ksteps_per_split;
    int qrows;        // rows of qweight (K/32*bits)
    unsigned long long kpg_inv;   // ceil(2^32 / (group_size / BK)): group of K-step kt = (kt * kpg_inv) >> 32, exact for kt < 2^16
    // balanced tail (gemm_kernel<..., TAIL = true>): the last `tail` logical tiles are run as 2^tail_lg K slices by as many workgroups each
    int tail, tail_lg;
    unsigned* tail_flags;         // workspace header, ticket half: [16 t] arrival ticket, [16 t + 1 + slice] "slice published"; zero before and after every launch
    float* tail_partial;          // [tail][2^tail_lg][32][256] float4: the accumulators of the slices that did not arrive last
    unsigned* err;                // sticky error word of the workspace header (a bounded wait gave up)
    unsigned max_spins;
};

__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ unsigned f16x2_bits(f16x2 v) { return __builtin_bit_cast(unsigned, v); }
// (a & mask) | orv in one VALU op (hipcc emits v_and + v_or for the C expression: VOP3 takes no literals on gfx9)
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask, unsigned orv) {
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
    return r;
}

template <typename T> struct Mma;
template <> struct Mma<f16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(a), as_f16x8(b), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ unsigned short t_bits(f16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ unsigned short t_bits(bf16 v) { return __builtin_bit_cast(unsigned short, v); }

// ---- per-lane weight words of one 16-deep MFMA k-step -----------------------------------------
// The lane's 8 k values start at bit B0 = (k0 + 8*half) * BITS of its column's bit stream (words down the
// column, N apart).  NW = words it has to fetch (per column) to cover 8*BITS bits from B0.
template <int BITS> struct BWords { static constexpr int n = (BITS == 3 || BITS == 8) ? 2 : 1; };

template <int BITS>
struct BRaw {                       // raw words of the lane's 2 columns for one k16 step
    u32x2 w[BWords<BITS>::n];
};

template <int BITS>
__device__ __forceinline__ void load_braw(BRaw<BITS>& r, const unsigned* __restrict__ qcol, int N, int qrows, int k) {
    const unsigned bit = (unsigned)k * BITS;
    const int wi = (int)(bit >> 5);
#pragma unroll
    for (int i = 0; i < BWords<BITS>::n; ++i) {
        const int row = min(wi + i, qrows - 1);     // the clamp only ever triggers for a word that is not used
        r.w[i] = *(const u32x2*)(qcol + (size_t)row * N);
    }
}

// 64-bit window of column c holding the lane's 8 fields starting at bit 0
template <int BITS>
__device__ __forceinline__ unsigned long long window(const BRaw<BITS>& r, int c, int k) {
    const unsigned sh = ((unsigned)k * BITS) & 31u;
    unsigned long long v = r.w[0][c];
    if constexpr (BWords<BITS>::n == 2) v |= (unsigned long long)r.w[1][c] << 32;
    return v >> sh;
}

// ---- per-column group constants ----------------------------------------------------------------
// 2 scales (T,T) ; the raw qzeros word(s) holding the lane's columns + the bit offset of the first one.  Nothing is computed on the
// loaded values at load time: a shift issued next to the load pins an s_waitcnt for a load that was requested one instruction ago
// in front of the step's MFMAs (seen in the two-K-group kernel: a full memory latency every other K-step).
struct CRaw { unsigned s; unsigned long long z; unsigned sh; };

template <int BITS>
__device__ __forceinline__ void load_craw(CRaw& c, const void* __restrict__ scales, const unsigned* __restrict__ qzeros,
                                          int g, int N, int n) {
    c.s = *(const unsigned*)((const unsigned short*)scales + (size_t)g * N + n);
    const int zrow_words = N / 32 * BITS;
    const unsigned bit = (unsigned)n * BITS;
    const int wi = (int)(bit >> 5);
    const unsigned sh = bit & 31u;
    const unsigned* zr = qzeros + (size_t)g * zrow_words;
    unsigned long long v = zr[wi];
    if constexpr (BITS == 3) v |= (unsigned long long)zr[min(wi + 1, zrow_words - 1)] << 32;
    c.z = v;
    c.sh = sh;
}

template <int BITS>
__device__ __forceinline__ int zero_point(const CRaw& c, int col, int zero_mode) {
    constexpr unsigned maxq = (1u << BITS) - 1u;
    const int f = (int)((unsigned)(c.z >> (c.sh + BITS * col)) & maxq) + 1;
    return zero_mode == GPTQ_ZERO_WRAP ? (f & (int)maxq) : f;
}

// ---- dequant of one word -> B fragment (4 regs, slot order k0,k4,k1,k5,k2,k6,k3,k7) -------------
template <int BITS, typename T> struct Deq {
    float s[2];
    int z[2];
    __device__ __forceinline__ void setup(const CRaw& c, int zero_mode) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const unsigned short sb = (unsigned short)(col ? (c.s >> 16) : (c.s & 0xffffu));
            s[col] = DType<T>::to_f32(__builtin_bit_cast(T, sb));
            z[col] = zero_point<BITS>(c, col, zero_mode);
        }
    }
    __device__ __forceinline__ u32x4 frag(const BRaw<BITS>& r, int col, int k) const {
        constexpr unsigned maxq = (1u << BITS) - 1u;
        const unsigned long long v = window<BITS>(r, col, k);
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = (int)((unsigned)(v >> (BITS * j)) & maxq);
            e[j] = t_bits(DType<T>::from_f32(s[col] * (float)(f - z[col])));   // exact product, one rounding
        }
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (unsigned)e[i] | ((unsigned)e[i + 4] << 16);
        return o;
    }
};

template <> struct Deq<4, f16> {
    f16x2 s2[2], c1[2], c2[2];
    __device__ __forceinline__ void setup(const CRaw& c, int zero_mode) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const unsigned sb = col ? (c.s >> 16) : (c.s & 0xffffu);
            s2[col] = as_f16x2(sb * 0x00010001u);
            const unsigned z = (unsigned)zero_point<4>(c, col, zero_mode);   // 0..16
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);               // -(1024 + z): 0xE400 = -1024, ulp 1
            const f16x2 k960 = {(f16)960.f, (f16)960.f};
            c2[col] = c1[col] + k960;                                        // -(64 + z), exact
        }
    }
    __device__ __forceinline__ u32x4 frag(const BRaw<4>& r, int col, int) const {
        const unsigned q = r.w[0][col], q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1[col];          // k0,k4 : w - z
        const f16x2 h1 = as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];    // k1,k5
        const f16x2 h2 = as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1[col];         // k2,k6
        const f16x2 h3 = as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];   // k3,k7
        u32x4 o;
        o[0] = f16x2_bits(h0 * s2[col]);
        o[1] = f16x2_bits(h1 * s2[col]);
        o[2] = f16x2_bits(h2 * s2[col]);
        o[3] = f16x2_bits(h3 * s2[col]);
        return o;
    }
};

// 8-bit and 3-bit fp16: the same exact packed form (csrc/gemv.hip, MagicF16, explains the arithmetic).  Both come out in the 4-bit slot
// order k0,k4,k1,k5,k2,k6,k3,k7, so the x staging is shared.  Plain C with literal masks on purpose -- no VALU instruction hidden in
// inline asm next to the matrix core (the hazard recognizer does not see it); hipcc spends v_and + v_or on it.
//   8-bit: the lane's 8 values are two words; v_perm joins their low halves (f0 f1 | f4 f5) and their high halves (f2 f3 | f6 f7), a
//          byte mask pairs (f0, f4), (f2, f6) and, shifted by 8, (f1, f5), (f3, f7): ~20 VALU per 8 weights instead of ~48.
//   3-bit: the lane's 8 values are a 24-bit window of the column's bit stream; low half = bits 0..15 (f0..f4 at 0, 3, 6, 9, 12), high
//          half = bits 12..27 (f4..f7 at 0, 3, 6, 9): pairs (f0, f4), (f1, f5), (f2, f6) at bits 0 / 3 / 6 and (f3, f7) at bit 3 of t >> 6.
template <> struct Deq<8, f16> {
    f16x2 s2[2], c1[2];
    __device__ __forceinline__ void setup(const CRaw& c, int zero_mode) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const unsigned sb = col ? (c.s >> 16) : (c.s & 0xffffu);
            s2[col] = as_f16x2(sb * 0x00010001u);
            const unsigned z = (unsigned)zero_point<8>(c, col, zero_mode);   // 0..256
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);               // -(1024 + z)
        }
    }
    __device__ __forceinline__ u32x4 frag(const BRaw<8>& r, int col, int) const {
        const unsigned w0 = r.w[0][col], w1 = r.w[1][col];                   // k0..k3, k4..k7
        const unsigned lo = __builtin_amdgcn_perm(w1, w0, 0x05040100u);      // f0 f1 | f4 f5
        const unsigned hi = __builtin_amdgcn_perm(w1, w0, 0x07060302u);      // f2 f3 | f6 f7
        const f16x2 h0 = as_f16x2((lo & 0x00ff00ffu) | 0x64006400u) + c1[col];            // k0,k4 : w - z
        const f16x2 h1 = as_f16x2(((lo >> 8) & 0x00ff00ffu) | 0x64006400u) + c1[col];     // k1,k5
        const f16x2 h2 = as_f16x2((hi & 0x00ff00ffu) | 0x64006400u) + c1[col];            // k2,k6
        const f16x2 h3 = as_f16x2(((hi >> 8) & 0x00ff00ffu) | 0x64006400u) + c1[col];     // k3,k7
        u32x4 o;
        o[0] = f16x2_bits(h0 * s2[col]);
        o[1] = f16x2_bits(h1 * s2[col]);
        o[2] = f16x2_bits(h2 * s2[col]);
        o[3] = f16x2_bits(h3 * s2[col]);
        return o;
    }
};
// 2-bit fp16 (round 3): the lane's 8 values are a 16-bit window; one v_perm spreads its bytes over the halves of a register (f0..f3 | f4..f7 at bits
// 0, 2, 4, 6), then (f, f + 4) pairs by v_and_or + packed fma with -(1024 * 2^-s + z): the 4-bit slot order, ~14 VALU per 8 weights instead of ~48
template <> struct Deq<2, f16> {
    f16x2 s2[2], c1[2], c2[2], c4[2], c6[2];
    __device__ __forceinline__ void setup(const CRaw& c, int zero_mode) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const unsigned sb = col ? (c.s >> 16) : (c.s & 0xffffu);
            s2[col] = as_f16x2(sb * 0x00010001u);
            const unsigned z = (unsigned)zero_point<2>(c, col, zero_mode);   // 0..4
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);               // -(1024 + z)
            const f16x2 k768 = {(f16)768.f, (f16)768.f}, k960 = {(f16)960.f, (f16)960.f}, k1008 = {(f16)1008.f, (f16)1008.f};
            c2[col] = c1[col] + k768;                                        // -(256 + z), exact
            c4[col] = c1[col] + k960;                                        // -(64 + z)
            c6[col] = c1[col] + k1008;                                       // -(16 + z)
        }
    }
    __device__ __forceinline__ u32x4 frag(const BRaw<2>& r, int col, int k) const {
        const unsigned v = (unsigned)window<2>(r, col, k);                   // 16 bits: f0..f7 at bits 2i
        const unsigned t = __builtin_amdgcn_perm(v, v, 0x0c010c00u);         // byte 0 -> bits 0..7, byte 1 -> bits 16..23
        const f16x2 r4 = {(f16)0.25f, (f16)0.25f}, r16 = {(f16)0.0625f, (f16)0.0625f}, r64 = {(f16)0.015625f, (f16)0.015625f};
        const f16x2 h0 = as_f16x2((t & 0x00030003u) | 0x64006400u) + c1[col];             // k0,k4 : w - z
        const f16x2 h1 = as_f16x2((t & 0x000C000Cu) | 0x64006400u) * r4 + c2[col];        // k1,k5
        const f16x2 h2 = as_f16x2((t & 0x00300030u) | 0x64006400u) * r16 + c4[col];       // k2,k6
        const f16x2 h3 = as_f16x2((t & 0x00C000C0u) | 0x64006400u) * r64 + c6[col];       // k3,k7
        u32x4 o;
        o[0] = f16x2_bits(h0 * s2[col]);
        o[1] = f16x2_bits(h1 * s2[col]);
        o[2] = f16x2_bits(h2 * s2[col]);
        o[3] = f16x2_bits(h3 * s2[col]);
        return o;
    }
};
template <> struct Deq<3, f16> {
    f16x2 s2[2], c1[2], c3[2], c6[2];
    __device__ __forceinline__ void setup(const CRaw& c, int zero_mode) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const unsigned sb = col ? (c.s >> 16) : (c.s & 0xffffu);
            s2[col] = as_f16x2(sb * 0x00010001u);
            const unsigned z = (unsigned)zero_point<3>(c, col, zero_mode);   // 0..8
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);               // -(1024 + z)
            const f16x2 k896 = {(f16)896.f, (f16)896.f}, k1008 = {(f16)1008.f, (f16)1008.f};
            c3[col] = c1[col] + k896;                                        // -(128 + z), exact
            c6[col] = c1[col] + k1008;                                       // -(16 + z), exact
        }
    }
    __device__ __forceinline__ u32x4 frag(const BRaw<3>& r, int col, int k) const {
        const unsigned v = (unsigned)window<3>(r, col, k);                   // 24 bits: f0..f7 at bits 3i
        const unsigned t = __builtin_amdgcn_perm(v >> 12, v, 0x05040100u);
        const unsigned t6 = t >> 6;
        const f16x2 r8 = {(f16)0.125f, (f16)0.125f}, r64 = {(f16)0.015625f, (f16)0.015625f};
        const f16x2 h0 = as_f16x2((t & 0x00070007u) | 0x64006400u) + c1[col];             // k0,k4 : w - z
        const f16x2 h1 = as_f16x2((t & 0x00380038u) | 0x64006400u) * r8 + c3[col];        // k1,k5
        const f16x2 h2 = as_f16x2((t & 0x01C001C0u) | 0x64006400u) * r64 + c6[col];       // k2,k6
        const f16x2 h3 = as_f16x2((t6 & 0x00380038u) | 0x64006400u) * r8 + c3[col];       // k3,k7
        u32x4 o;
        o[0] = f16x2_bits(h0 * s2[col]);
        o[1] = f16x2_bits(h1 * s2[col]);
        o[2] = f16x2_bits(h2 * s2[col]);
        o[3] = f16x2_bits(h3 * s2[col]);
        return o;
    }
};

// 4-bit bf16: w - z is formed exactly with the fp16 magic numbers (gfx950 has no packed bf16 arithmetic), each value is
// multiplied by the scale in fp32 (exact product) and rounded once to bf16 by v_cvt_pk_bf16_f32 -- the reference's
// scales * (weight - zeros) in bf16, bit for bit -- at about 21-29 VALU per word instead of ~40 for the per-field path.
template <> struct Deq<4, bf16> {
    f16x2 c1[2], c2[2];
    float s[2];
    __device__ __forceinline__ void setup(const CRaw& c, int zero_mode) {
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const unsigned short sb = (unsigned short)(col ? (c.s >> 16) : (c.s & 0xffffu));
            s[col] = (float)__builtin_bit_cast(bf16, sb);
            const unsigned z = (unsigned)zero_point<4>(c, col, zero_mode);
            c1[col] = as_f16x2(z * 0x00010001u + 0xE400E400u);
            const f16x2 k960 = {(f16)960.f, (f16)960.f};
            c2[col] = c1[col] + k960;
        }
    }
    static __device__ __forceinline__ unsigned scaled_pair(f16x2 h, float sc) {
        // v_fma_mix_f32 reads either half of the packed fp16 register directly (fp32 result): no separate v_cvt_f32_f16
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        float lo, hi;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(sc));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(hb), "v"(sc));
        const bf16x2 v = {(bf16)lo, (bf16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ __forceinline__ u32x4 frag(const BRaw<4>& r, int col, int) const {
        const unsigned q = r.w[0][col], q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1[col];
        const f16x2 h1 = as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];
        const f16x2 h2 = as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1[col];
        const f16x2 h3 = as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2[col];
        u32x4 o;
        o[0] = scaled_pair(h0, s[col]);
        o[1] = scaled_pair(h1, s[col]);
        o[2] = scaled_pair(h2, s[col]);
        o[3] = scaled_pair(h3, s[col]);
        return o;
    }
};

// ---- the kernel -------------------------------------------------------------------------------
// Workgroup = 4 waves side by side along N: tile BM = 32*MT rows x 256 columns, K-step BK.
// VAR selects the inner-loop schedule: 1 (default) = explicit software pipeline over the MFMA k-steps, 0 = plain loop
// scheduled by hipcc, 2 = plain + s_setprio around the MFMA groups, 3 = the pipeline carried across K-steps (+-1 %: kept as
// an experiment, tuning.reserved[3] = 3).  Within-run A/B on MI355X (tools/gemmlab, min of 3
// rounds, TFLOP/s at M=2048 4096^2 / M=4096 4096^2 / 4096x11008 / 11008x4096): VAR1 793/1002/908/889, VAR0 678/990/888/840,
// VAR2 743/972/852/815.  (Also tried: 8 waves per workgroup, one column of each pair per wave -- slower everywhere.)
// XPRE: x already arrives in k-slot order (act-order layers: the column-permute pre-pass writes it that way).
// GLDS (needs XPRE, BK = 64): x goes global -> LDS by DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write, no
// v_perm.  The LDS image of a wave's 64 lanes is linear (8 rows x 128 B), so rows are unpadded and the bank spread comes
// from an XOR swizzle applied to the SOURCE chunk: LDS slot s of row r holds k-chunk s ^ ((r >> 1) & 7); the A-fragment
// reads apply the same XOR (16 distinct 16-byte units per ds_read_b128 lane group -> conflict free).
// KG = 2 (needs MT = 4): 8 waves; waves 4-7 are a second copy of the 1x4 arrangement working on the other half of the
// block's K range with their own x buffers, and the two halves are summed through LDS at the end.  Used when the launch
// has at most one 128x256 tile per CU: a CU then holds two waves per SIMD (what two co-resident workgroups would give a
// larger problem), so one wave's dequant/LDS work fills the other's MFMA shadows, at the same weight/x traffic per flop.
// TAIL (round 3, KG = 1, one K slice): balanced tail.  The workgroups of a launch start together and take the same time, so T tiles cost about
// ceil(T / 256) tile times: the T mod 256 tiles of the last round keep a few CUs busy while the rest of the chip idles (measured: 4096x11008
// at M = 1536, 516 tiles, 179 us against 137 us of work; still 5 % at 1376 tiles).  With TAIL the last p.tail = T mod 256 logical
// tiles are run by 2^tail_lg workgroups each, one per K slice, launched behind the whole tiles.  Every slice takes an arrival ticket when its K
// loop is done; all but the last arrival write their accumulators to the workspace (write-through stores), raise their "published" flag and
// leave; the last arrival -- which only ever waits for workgroups that are past their K loops: no residency assumption -- adds the slices IN
// SLICE ORDER (its own from registers), so the result does not depend on who arrives last, and stores the tile.  It also clears the words
// (the header stays zero between launches).  The fix-up moves 128 KiB per slice and direction, so it only pays for small tails: plan_gemm.
template <int BITS, typename T, int MT, int BK, int VAR = 1, bool XPRE = false, bool GLDS = false, int KG = 1, bool TAIL = false>
__global__ void __launch_bounds__(256 * KG, 2 / KG) gemm_kernel(GemmParams p) {
    static_assert(!TAIL || (KG == 1 && MT == 4), "the balanced tail is built for the one-K-group 128-row tile");
    constexpr int KS = BK / 16;                // MFMA k-steps per K-step
    constexpr int BM = 32 * MT;
    static_assert(!GLDS || (XPRE && BK == 64), "the DMA staging needs pre-slotted x and 128-byte rows");
    constexpr int STRIDE = GLDS ? BK * 2 : BK * 2 + 16;   // bytes per LDS row of x (padded unless DMA-staged)
    constexpr int CPR = BK / 8;                // 16-byte chunks per row
    constexpr int CHUNKS = BM * CPR;
    constexpr int NTHR = 256;
    constexpr int NCH = (CHUNKS + NTHR - 1) / NTHR;
    static_assert(KG == 1 || (KG == 2 && MT == 4), "K groups: 1, or 2 with the 128-row tile");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];   // KG x 2 x BM x STRIDE
    // wave-uniform by construction; said so, or every buffer load whose scalar offset depends on the K group turns into a waterfall loop
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    char* const smem = smem_all + (size_t)kg * (2 * BM * STRIDE);

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // Logical tile order: column blocks 8 tiles wide, walked row by row, so the contiguous run of logical ids
    // an XCD receives is a compact (rows x 8 columns) patch: its L2 holds 8 weight panels and a few x panels
    // instead of every x panel (measured: L2 fill traffic was 0.49 GB per 4096^3 launch with a column-major order).
    int L, tail_half = -1;                     // tail_half >= 0: this workgroup runs K slice tail_half of tail tile L (wave-uniform: blockIdx only)
    if constexpr (TAIL) {
        const int whole = p.nbm * p.nbn - p.tail;
        if ((int)blockIdx.x < whole) L = xcd_remap(blockIdx.x, whole);
        else {
            const int j = xcd_remap((int)blockIdx.x - whole, p.tail << p.tail_lg);      // the slices of a tile are neighbours (one XCD when the tail is a multiple of 8; speed only)
            L = whole + (j >> p.tail_lg);
            tail_half = j & ((1 << p.tail_lg) - 1);
        }
    } else L = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    int bm, bn;
    {
        const int full = p.nbn >> 3, per = p.nbm * 8;
        if (L < full * per) {
            const int cb = L / per, r = L - cb * per;
            bm = r >> 3;
            bn = cb * 8 + (r & 7);
        } else {
            const int w = p.nbn & 7, r = L - full * per;
            bm = r / w;
            bn = full * 8 + (r - bm * w);
        }
    }
    const int m0 = bm * BM;
    const int n = bn * 256 + wave * 64 + 2 * l31;      // this lane's first column (second is n + 1)
    const bool col_ok = n < p.N;
    const int nl = col_ok ? n : 0;
    const unsigned* __restrict__ qcol = p.qweight + nl;
    const unsigned short* __restrict__ x = (const unsigned short*)p.x;

    int kt0 = blockIdx.y * p.ksteps_per_split;
    int kt1 = min(kt0 + p.ksteps_per_split, p.ksteps_total);
    if constexpr (KG == 2) {                   // the planner only picks KG = 2 when every slice has an even step count
        const int hs = (kt1 - kt0) >> 1;
        kt0 += kg * hs;
        kt1 = kt0 + hs;
    }
    if constexpr (TAIL) {
        if (tail_half >= 0) {                  // the planner only cuts K into slices of equal, whole step counts
            const int hs = p.ksteps_total >> p.tail_lg;
            kt0 = tail_half * hs;
            kt1 = kt0 + hs;
        }
    }

    // A staging assignment: chunk c -> (row, 16-byte column)
    // Rows past M are clamped to row M-1 (never predicated: a predicated load splits the K-loop into several
    // basic blocks and the prefetch stops overlapping the MFMAs); their accumulators are simply not stored.
    int a_row[NCH], a_kc[NCH];
    unsigned a_off[NCH];
    const unsigned short* a_src[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NTHR;
        a_row[i] = c / CPR;
        a_kc[i] = c - a_row[i] * CPR;
        const int src_kc = GLDS ? (a_kc[i] ^ ((a_row[i] >> 1) & 7)) : a_kc[i];
        a_src[i] = x + (size_t)min(m0 + a_row[i], p.M - 1) * p.K + src_kc * 8;
        a_off[i] = (unsigned)(min(m0 + a_row[i], p.M - 1) - m0) * (unsigned)p.K * 2u + (unsigned)src_kc * 16u;   // < 2^32: BM rows
    }
    const char* a_base = (const char*)(x + (size_t)m0 * p.K);             // uniform
    // Buffer descriptors (wave-uniform: kernel arguments and blockIdx only) -- buffer_load takes (SGPR descriptor, VGPR 32-bit
    // offset, SGPR offset), so the per-K-step part of every address is scalar arithmetic.
    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qweight, 0, (int)((size_t)p.qrows * p.N * 4), 0x00020000);
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)((size_t)BM * p.K * 2), 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, 0, (int)((size_t)(p.K / p.group_size) * p.N * 2), 0x00020000);
    const int zrow_bytes = p.N / 32 * BITS * 4;
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.qzeros, 0, (p.K / p.group_size) * zrow_bytes, 0x00020000);
    // DMA: instruction i of wave w fills LDS chunks [(i*4 + w)*64, +64) = 1 KiB, lane l -> chunk base + l (= tid + i*256)
    auto dma_a = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            char* dst = smem + (size_t)buf * (BM * STRIDE) + (size_t)(i * NTHR + wave * 64) * 16;
            lds_dma16(a_base + (size_t)kt * (BK * 2) + a_off[i], lds_addr_of(dst));
        }
    };
    auto load_a = [&](int kt, u32x4 (&r)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, a_off[i], (unsigned)kt * (BK * 2), 0);
    };
    auto store_a = [&](int buf, const u32x4 (&r)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (CHUNKS % NTHR != 0 && tid + i * NTHR >= CHUNKS) continue;
            // (x0,x1)(x2,x3)(x4,x5)(x6,x7) -> (x0,x4)(x1,x5)(x2,x6)(x3,x7): the slot order of the B fragments
            u32x4 o;
            if constexpr (XPRE) {
                o = r[i];
            } else {
                o[0] = __builtin_amdgcn_perm(r[i][2], r[i][0], 0x05040100u);
                o[1] = __builtin_amdgcn_perm(r[i][2], r[i][0], 0x07060302u);
                o[2] = __builtin_amdgcn_perm(r[i][3], r[i][1], 0x05040100u);
                o[3] = __builtin_amdgcn_perm(r[i][3], r[i][1], 0x07060302u);
            }
            *(u32x4*)(smem + (size_t)buf * (BM * STRIDE) + a_row[i] * STRIDE + a_kc[i] * 16) = o;
        }
    };
    // Addresses are split into a wave-uniform part (K-step / k-step: scalar ALU) and a loop-invariant 32-bit per-lane byte
    // offset, so a load costs no vector address arithmetic inside the K loop (global_load with an SGPR base).
    const unsigned b_lane_off = ((unsigned)nl + (unsigned)half * (BITS == 8 ? 2u : 1u) * (unsigned)p.N) * 4u;
    const unsigned s_lane_off = (unsigned)nl * 2u;
    const unsigned z_lane_off = (((unsigned)nl * BITS) >> 5) * 4u, z_lane_sh = ((unsigned)nl * BITS) & 31u;
    auto load_b = [&](int kt, BRaw<BITS> (&b)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (BITS == 4 || BITS == 8) {
                constexpr int WPH = (BITS == 8) ? 2 : 1;                       // words per 8 k
                const size_t row_u = (size_t)(kt * (BK / 8) + ks * 2) * WPH;      // uniform packed row of lane-half 0
#pragma unroll
                for (int i = 0; i < WPH; ++i)
                    b[ks].w[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_q, b_lane_off, (unsigned)((row_u + i) * (size_t)p.N * 4), 0);
            } else {
                load_braw<BITS>(b[ks], qcol, p.N, p.qrows, kt * BK + ks * 16 + half * 8);
            }
        }
    };
    auto load_c = [&](int kt, CRaw& c) {
        const int g = (int)(((unsigned long long)(unsigned)kt * p.kpg_inv) >> 32);
        if constexpr (BITS != 3) {
            c.s = __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, s_lane_off, (unsigned)g * (unsigned)p.N * 2u, 0);
            c.z = (unsigned long long)__builtin_amdgcn_raw_buffer_load_b32(rsrc_z, z_lane_off, (unsigned)(g * zrow_bytes), 0);
            c.sh = z_lane_sh;
        } else {
            load_craw<BITS>(c, p.scales, p.qzeros, g, p.N, nl);
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // VAR == 16 (tools/gemmlab only, -DGPTQ_GEMM_ABLATIONS): the default schedule with s_memtime stamps at the phase boundaries of every
    // K-step; per wave the cycle sums of the five phases go to p.partial[wave * 8 ..] of workgroup 0 (the stamps cost ~10 % themselves).
    constexpr bool STAMP = (VAR == 16);
    // VAR == 24 (tools/gemmlab only): the next step's loads are not issued in a block at the top of the step but between the MFMA groups
    // (weights + constants behind group 0, x behind group 1, the x tile goes to LDS behind group 2) -- what the timeline suggests
    constexpr bool ILV = (VAR == 24);
    // VAR == 32 (tools/gemmlab only, KG = 2): PING-PONG between the two K groups.  Waves w (group 0) and w + 4 (group 1) share a SIMD; a token
    // word per pair in LDS says whose turn the MFMA half of a step is, so one wave's load / unpack half runs under the other's MFMA half instead of
    // both sitting in the same phase (the K-step timeline).  The per-step barrier becomes per group (an LDS counter): the groups share
    // nothing until the final reduction.  Every spin is bounded (a lost token ends in wrong numbers, never in a hung queue).
    constexpr bool PP = (VAR == 32) && KG == 2;
    unsigned* const pp_sync = (unsigned*)(smem_all + (size_t)KG * 2 * BM * STRIDE);      // [0..3] pair tokens, [4..5] group barrier counters
    [[maybe_unused]] unsigned pp_step = 0, pp_epoch = 0;
    if constexpr (PP) {
        if (threadIdx.x < 8) pp_sync[threadIdx.x] = 0u;                                 // published by the prologue barrier below
    }
    auto pp_spin_until = [&](const unsigned* w, unsigned want) {                          // wave-uniform LDS poll, bounded
        __builtin_amdgcn_sched_barrier(0);
        for (int spins = 0; spins < (1 << 16); ++spins) {
            if ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    [[maybe_unused]] unsigned tsum[5] = {0u, 0u, 0u, 0u, 0u};
    auto clk = [] { __builtin_amdgcn_sched_barrier(0); const unsigned t = (unsigned)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); return t; };

    // Two K-steps per loop trip, ping-ponging between register sets (b0,c0)/(b1,c1) and LDS buffers 0/1, so
    // nothing is copied and every LDS offset is an immediate.  Inside a step the A fragments of MFMA k-step
    // ks+1 are read from LDS before the MFMAs of ks are issued (their latency hides behind 8 MFMAs).
    u32x4 a_next[NCH];
    BRaw<BITS> b0[KS], b1[KS];
    CRaw c0, c1;
    if constexpr (GLDS) dma_a(kt0, 0); else load_a(kt0, a_next);
    load_b(kt0, b0);
    load_c(kt0, c0);
    if constexpr (!GLDS) store_a(0, a_next);
    if constexpr (GLDS) wait_vmcnt<0>();           // the DMA is invisible to the compiler's own waits
    __syncthreads();

    const int a_lane_off = GLDS ? l31 * STRIDE : l31 * STRIDE + half * 16;
    const int a_swz = (l31 >> 1) & 7;                  // GLDS: XOR applied to the 16-byte slot index
    // VAR == 3: fragments of the NEXT K-step's first MFMA k-step, produced under the last MFMAs of the current K-step
    u32x4 a_first[MT], bq_first[2];
    Deq<BITS, T> dq_cur;
    auto read_a = [&](int buf, int ks, u32x4 (&dst)[MT]) {
        const char* base = smem + buf * (BM * STRIDE) + a_lane_off;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            dst[mt] = *(const u32x4*)(base + mt * 32 * STRIDE + (GLDS ? (((ks * 2 + half) ^ a_swz) * 16) : ks * 32));
    };
    if constexpr (VAR == 3) {
        dq_cur.setup(c0, p.zero_mode);
        read_a(0, 0, a_first);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bq_first[nt] = dq_cur.frag(b0[0], nt, kt0 * BK + half * 8);
    }
    auto step = [&](int kt, auto bufc, const BRaw<BITS> (&b_use)[KS], const CRaw& c_use, BRaw<BITS> (&b_fill)[KS], CRaw& c_fill) {
        constexpr int BUF = decltype(bufc)::value;
        const int ktn = min(kt + 1, kt1 - 1);          // last step re-loads itself (no branch in the pipeline)
        [[maybe_unused]] unsigned T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0;
        if constexpr (STAMP) T0 = clk();
        if constexpr (GLDS) {
            // The DMAs below are invisible to the compiler's vmcnt bookkeeping; any wait it emits while they are in flight is too
            // strict by their number (in-order retirement).  So this step's weight words and group constants -- requested a whole
            // step ago -- are claimed HERE, before anything new is issued: the compiler's exact wait lands on this line, and no
            // wait of its own follows while the DMAs fly.
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < BWords<BITS>::n; ++i) asm volatile("" ::"v"(b_use[ks].w[i][0]), "v"(b_use[ks].w[i][1]));
            asm volatile("" ::"v"(c_use.s), "v"((unsigned)c_use.z));
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!ILV) {
            if constexpr (GLDS) dma_a(ktn, BUF ^ 1);
            else if constexpr (!(VAR >= 8 && (VAR & 2))) load_a(ktn, a_next);
            load_b(ktn, b_fill);
            load_c(ktn, c_fill);
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ahead of this step's MFMAs
        if constexpr (STAMP) T1 = clk();

        if constexpr (VAR == 3) {
            // One continuous software pipeline across K-steps: the barrier sits in front of the LAST MFMA group of the
            // step (its operands are already in registers), and right behind it the first A fragments of the next step
            // are read and its first B fragments dequantised -- so the 8 MFMAs of that group cover the barrier wait, the
            // LDS latency and the dequant, and a K-step starts with its MFMAs instead of a fragment-production bubble.
            u32x4 a[2][MT], bq[2][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[0][mt] = a_first[mt];
            bq[0][0] = bq_first[0];
            bq[0][1] = bq_first[1];
            Deq<BITS, T> dq_nx;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                    read_a(BUF, ks + 1, a[(ks + 1) & 1]);
                } else {
                    if constexpr (!GLDS) store_a(BUF ^ 1, a_next);
                    if constexpr (GLDS) wait_vmcnt<KS * (BITS == 8 ? 2 : 1) + 2>();     // this step's DMAs (see the end of step())
                    __syncthreads();
                    read_a(BUF ^ 1, 0, a_first);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < KS) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) bq[(ks + 1) & 1][nt] = dq_cur.frag(b_use[ks + 1], nt, kt * BK + (ks + 1) * 16 + half * 8);
                } else {
                    dq_nx.setup(c_fill, p.zero_mode);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) bq_first[nt] = dq_nx.frag(b_fill[0], nt, ktn * BK + half * 8);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Mma<T>::run(a[ks & 1][mt], bq[ks & 1][nt], acc[mt][nt]);
            }
            dq_cur = dq_nx;
            return;
        }
        Deq<BITS, T> dq;
        dq.setup(c_use, p.zero_mode);
        const char* abase = smem + BUF * (BM * STRIDE) + a_lane_off;
        if constexpr (VAR == 1 || VAR >= 8) {
            // (VAR >= 8: timing ablations of this schedule -- bit 0 skips the dequant math, bit 1 the x staging,
            //  bit 2 the barrier; their results are wrong by construction and only tools/gemmlab selects them)
            auto frag = [&](const BRaw<BITS>& r, int col, int k) -> u32x4 {
                if constexpr (VAR >= 8 && (VAR & 1)) { const unsigned q = r.w[0][col]; return u32x4{q, q ^ 0x11111111u, q ^ 0x22222222u, q ^ 0x44444444u}; }
                else return dq.frag(r, col, k);
            };
            // explicit software pipeline over the KS MFMA k-steps: the fragments of ks+1 (A from LDS, B dequantised in
            // registers) are produced while the 2*MT MFMAs of ks run; sched_barrier pins the LDS reads at the top of
            // each region (hipcc otherwise sinks them right in front of their first use and exposes the LDS latency)
            u32x4 a[2][MT], bq[2][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[0][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (GLDS ? ((half ^ a_swz) * 16) : 0));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[0][nt] = frag(b_use[0], nt, kt * BK + half * 8);
            if constexpr (STAMP) {                         // first fragments in registers: this step's weight words and LDS tile have arrived
                asm volatile("" ::"v"(a[0][0][0]), "v"(bq[0][0][0]), "v"(bq[0][1][3]));
                T2 = clk();
            }
            if constexpr (PP) {                            // my turn on the matrix core?  (group 0 owns the even half steps)
                asm volatile("" ::"v"(a[0][0][0]), "v"(bq[0][0][0]), "v"(bq[0][1][3]));
                pp_spin_until(pp_sync + wave, 2u * pp_step + (unsigned)kg);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[(ks + 1) & 1][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (GLDS ? ((((ks + 1) * 2 + half) ^ a_swz) * 16) : (ks + 1) * 32));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < KS) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) bq[(ks + 1) & 1][nt] = frag(b_use[ks + 1], nt, kt * BK + (ks + 1) * 16 + half * 8);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Mma<T>::run(a[ks & 1][mt], bq[ks & 1][nt], acc[mt][nt]);
                if constexpr (ILV && !GLDS) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) { load_b(ktn, b_fill); load_c(ktn, c_fill); }
                    if (ks == 1) load_a(ktn, a_next);
                    if (ks == KS - 2 && KS >= 4) store_a(BUF ^ 1, a_next);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 a[MT], b[2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (GLDS ? (((ks * 2 + half) ^ a_swz) * 16) : ks * 32));
                const int k = kt * BK + ks * 16 + half * 8;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) b[nt] = dq.frag(b_use[ks], nt, k);
                if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Mma<T>::run(a[mt], b[nt], acc[mt][nt]);
                if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(0);
            }
        }
        if constexpr (PP) {                            // all MFMAs of the step issued: the sibling wave on this SIMD may start its own
            __builtin_amdgcn_sched_barrier(0);
            if (lane == 0) __hip_atomic_store(pp_sync + wave, 2u * pp_step + (unsigned)kg + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++pp_step;
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (STAMP) T3 = clk();               // all MFMAs of the step issued
        if constexpr (!GLDS && !(VAR >= 8 && (VAR & 2)) && !ILV) store_a(BUF ^ 1, a_next);
        // DMA-staged x: the next step's tile must have landed before anybody passes the barrier.  vmcnt retires in order and
        // the step issued, after its DMAs, KS * (1 or 2) weight loads + 2 group-constant loads: those may stay in flight.
        if constexpr (GLDS) wait_vmcnt<KS * (BITS == 8 ? 2 : 1) + 2>();
        if constexpr (STAMP) T4 = clk();               // next x tile written to LDS
        if constexpr (PP) {                            // per-group barrier: this group's 4 waves only (LDS operations of a wave execute in order, so
            ++pp_epoch;                                    // the counter increment is behind this wave's x-tile writes)
            if (lane == 0) __hip_atomic_fetch_add(pp_sync + 4 + kg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pp_spin_until(pp_sync + 4 + kg, 4u * pp_epoch);
        } else if constexpr (!(VAR >= 8 && (VAR & 4))) __syncthreads();
        if constexpr (STAMP) {
            const unsigned T5 = clk();                     // barrier released
            tsum[0] += T1 - T0; tsum[1] += T2 - T1; tsum[2] += T3 - T2; tsum[3] += T4 - T3; tsum[4] += T5 - T4;
        }
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
        step(kt, std::integral_constant<int, 0>{}, b0, c0, b1, c1);
        if (kt + 1 < kt1) step(kt + 1, std::integral_constant<int, 1>{}, b1, c1, b0, c0);
    }

    if constexpr (STAMP) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && p.partial != nullptr) {
            unsigned* o = (unsigned*)p.partial + (threadIdx.x >> 6) * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) o[i] = tsum[i];
            o[5] = (unsigned)(kt1 - kt0);
        }
    }
    if constexpr (PP) __syncthreads();                   // the groups ran unsynchronised: both must be out of their K loops before the exchange area is written
    if constexpr (KG == 2) {
        // sum the two K halves through LDS (the x buffers are dead after the last barrier): group 1 hands rows 0-63 to
        // group 0, then group 0 hands rows 64-127 to group 1; each group stores the half it completed.
        float4* ex = (float4*)smem_all;        // [(mt, nt, quad)][256 threads] float4: lane-contiguous, conflict free
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (kg != pass) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x16& a = acc[pass * 2 + mt][nt];
                            ex[((mt * 2 + nt) * 4 + q) * 256 + tid] = float4{a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
                        }
            }
            __syncthreads();
            if (kg == pass) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = ex[((mt * 2 + nt) * 4 + q) * 256 + tid];
                            f32x16& a = acc[pass * 2 + mt][nt];
                            a[q * 4] += v.x; a[q * 4 + 1] += v.y; a[q * 4 + 2] += v.z; a[q * 4 + 3] += v.w;
                        }
            }
            if (pass == 0) __syncthreads();
        }
    }

    if constexpr (TAIL) {
        if (tail_half >= 0) {
            const int t = L - (p.nbm * p.nbn - p.tail), nsl = 1 << p.tail_lg;
            unsigned* const tk = p.tail_flags + 16 * t;                            // [0] arrival ticket, [1 + slice] "slice published"
            constexpr size_t SLAB = 32 * 256 * 4;                                  // floats: [(mt, nt, quad)][256 threads] float4, lane-contiguous
            float* const slabs = p.tail_partial + ((size_t)t << p.tail_lg) * SLAB;
            unsigned* const lw = (unsigned*)smem_all;                              // the x buffers are dead behind the last step's barrier
            if (threadIdx.x == 0) lw[0] = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned arrival = (unsigned)__builtin_amdgcn_readfirstlane((int)lw[0]);
            if (arrival + 1u < (unsigned)nsl) {    // not the last: publish, drain the write-through stores, count, leave
                float* const mine = slabs + (size_t)tail_half * SLAB;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x16& a = acc[mt][nt];
                            const f32x4 v = {a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
                            // s_nop inside the string: hipcc pads nothing behind an asm statement, and the next instruction may overwrite the data
                            // registers (dead to the compiler) while the store still reads them -- measured: the first 8 bytes of lanes 12..15 of
                            // every 16 came out as the NEXT store's address (tools/tail_diag.py)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine + ((size_t)((mt * 2 + nt) * 4 + q) * 256 + tid) * 4), "v"(v) : "memory");
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every publishing wave drains its write-through stores
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(tk + 1 + tail_half, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if ((int)threadIdx.x < nsl && (int)threadIdx.x != tail_half) {      // last arrival: the others are past their K loops -- short, bounded waits
                unsigned* const f = tk + 1 + threadIdx.x;
                for (unsigned spins = 0;; ++spins) {
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    if (spins > p.max_spins) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            // sums in slice order, straight to the output (the accumulators are only read: a second, modified copy of 128 registers would spill)
            float bias0 = 0.f, bias1 = 0.f;
            if (p.bias && col_ok) {
                bias0 = DType<T>::to_f32(((const T*)p.bias)[n]);
                bias1 = DType<T>::to_f32(((const T*)p.bias)[n + 1]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s0[16], s1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
                for (int sl = 0; sl < nsl; ++sl) {
                    if (sl == tail_half) {             // wave-uniform
#pragma unroll
                        for (int r = 0; r < 16; ++r) { s0[r] += acc[mt][0][r]; s1[r] += acc[mt][1][r]; }
                    } else {
                        const float* from = slabs + (size_t)sl * SLAB;
                        unsigned long long pv[2][4][2];
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {      // agent-scope loads: they bypass this XCD's non-coherent L2 lines
                                const unsigned long long* src = (const unsigned long long*)(from + ((size_t)((mt * 2 + nt) * 4 + q) * 256 + tid) * 4);
                                pv[nt][q][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                pv[nt][q][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned long long w0 = pv[0][r >> 2][(r >> 1) & 1], w1 = pv[1][r >> 2][(r >> 1) & 1];
                            s0[r] += __builtin_bit_cast(float, (unsigned)((r & 1) ? (w0 >> 32) : w0));
                            s1[r] += __builtin_bit_cast(float, (unsigned)((r & 1) ? (w1 >> 32) : w1));
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M && col_ok) {
                        const unsigned o = (unsigned)t_bits(DType<T>::from_f32(s0[r] + bias0)) | ((unsigned)t_bits(DType<T>::from_f32(s1[r] + bias1)) << 16);
                        *(unsigned*)((unsigned short*)p.out + (size_t)m * p.N + n) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (!col_ok) return;
    float bias0 = 0.f, bias1 = 0.f;
    if (p.bias && p.ksplit == 1) {
        bias0 = DType<T>::to_f32(((const T*)p.bias)[n]);
        bias1 = DType<T>::to_f32(((const T*)p.bias)[n + 1]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (KG == 2 && (mt >> 1) != kg) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= p.M) continue;
            const float v0 = acc[mt][0][r], v1 = acc[mt][1][r];
            if (p.ksplit > 1) {
                float2 o = {v0, v1};
                *(float2*)(p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + n) = o;
            } else {
                const unsigned o = (unsigned)t_bits(DType<T>::from_f32(v0 + bias0)) |
                                   ((unsigned)t_bits(DType<T>::from_f32(v1 + bias1)) << 16);
                *(unsigned*)((unsigned short*)p.out + (size_t)m * p.N + n) = o;
            }
        }
    }
}

// ---- skinny variant: 8 < M <= 128 (weight-streaming regime) --------------------------------------
// Same fragment scheme (weights straight into B fragments), different decomposition: the work is bandwidth
// bound on the packed weights, so it is cut like the GEMV -- a workgroup owns ONE 64-column strip and its
// W waves split K (contiguous runs of 16-deep steps), which gives N/64 x ksplit workgroups of wide (256 B)
// row segments.  x is tiny and L2 resident, so A fragments are loaded straight from global (no LDS staging,
// no barrier in the K loop); the only synchronisation is the final cross-wave sum through LDS.
template <int BITS, typename T, int MT, int UNR>
__global__ void __launch_bounds__(512) gemm_skinny_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // W x 8 KiB
    float* red = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = blockIdx.z * (32 * MT);
    const int n = strip * 64 + 2 * l31;
    const bool col_ok = n < p.N;
    const int nl = col_ok ? n : 0;
    const unsigned* __restrict__ qcol = p.qweight + nl;
    const unsigned short* __restrict__ x = (const unsigned short*)p.x;
    const int S = p.K >> 4;                                       // 16-deep steps in K
    const int b0 = blockIdx.y * p.ksteps_per_split;               // this workgroup's steps [b0, b1)
    const int b1 = min(b0 + p.ksteps_per_split, S);
    const int spw = ((p.ksteps_per_split + W - 1) / W + UNR - 1) / UNR * UNR;
    const int ws = b0 + wave * spw, we = min(ws + spw, b1);

    const unsigned short* a_src[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_src[mt] = x + (size_t)min(m0 + mt * 32 + l31, p.M - 1) * p.K + half * 8;

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int s0 = ws; s0 < we; s0 += UNR) {
        // the UNR steps of a chunk share one group (group_size % (16*UNR) == 0, chunks are UNR-aligned)
        CRaw craw;
        load_craw<BITS>(craw, p.scales, p.qzeros, (s0 * 16) / p.group_size, p.N, nl);
        u32x4 a[UNR][MT];
        BRaw<BITS> braw[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int sj = min(s0 + j, we - 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[j][mt] = *(const u32x4*)(a_src[mt] + (size_t)sj * 16);
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) load_braw<BITS>(braw[j], qcol, p.N, p.qrows, min(s0 + j, we - 1) * 16 + half * 8);
        Deq<BITS, T> dq;
        dq.setup(craw, p.zero_mode);
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const bool live = s0 + j < we;
            const int k = min(s0 + j, we - 1) * 16 + half * 8;
            u32x4 b[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b[nt] = dq.frag(braw[j], nt, k);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const u32x4 t = a[j][mt];
                u32x4 o;
                o[0] = __builtin_amdgcn_perm(t[2], t[0], 0x05040100u);
                o[1] = __builtin_amdgcn_perm(t[2], t[0], 0x07060302u);
                o[2] = __builtin_amdgcn_perm(t[3], t[1], 0x05040100u);
                o[3] = __builtin_amdgcn_perm(t[3], t[1], 0x07060302u);
                if (!live) o = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Mma<T>::run(o, b[nt], acc[mt][nt]);
            }
        }
    }

    // ---- cross-wave sum, one 32-row tile at a time (W x 8 KiB of LDS) -------------------------------
    float bias0 = 0.f, bias1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt) __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * 2 + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
        __syncthreads();
        for (int e = tid; e < 16 * 64; e += blockDim.x) {        // e = r * 64 + lane'
            const int r = e >> 6, ln = e & 63;
            float v0 = 0.f, v1 = 0.f;
            for (int w = 0; w < W; ++w) {
                v0 += red[((w * 2 + 0) * 16 + r) * 64 + ln];
                v1 += red[((w * 2 + 1) * 16 + r) * 64 + ln];
            }
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
            const int nn = strip * 64 + 2 * (ln & 31);
            if (m >= p.M || nn >= p.N) continue;
            if (p.ksplit > 1) {
                float2 o = {v0, v1};
                *(float2*)(p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + nn) = o;
            } else {
                if (p.bias) {
                    bias0 = DType<T>::to_f32(((const T*)p.bias)[nn]);
                    bias1 = DType<T>::to_f32(((const T*)p.bias)[nn + 1]);
                }
                const unsigned o = (unsigned)t_bits(DType<T>::from_f32(v0 + bias0)) |
                                   ((unsigned)t_bits(DType<T>::from_f32(v1 + bias1)) << 16);
                *(unsigned*)((unsigned short*)p.out + (size_t)m * p.N + nn) = o;
            }
        }
    }
}

// ---- 16-column strips for 8 < M <= 64, 4-bit fp16 / bf16 ("batched decode") ----------------------------------------
// The decode decomposition (N/16 strips x 16 waves splitting K, no second launch for any Llama width) carried over to the
// matrix core that fits it: v_mfma_f32_16x16x32 takes, in lane l, the 8 consecutive k  8*(l>>4)..+7  of column l&15 of B --
// again exactly one 4-bit word -- and of row l&15 of A.  A wave-load of weights is 4 packed rows x 64 B; x fragments are
// 16-byte loads from L2 (x is M*K*2 bytes, shared by all strips).  Everything a wave needs for U k-steps is requested
// before the first MFMA.  RT row tiles of 16 reuse every dequantised word.  Cross-wave sum through LDS as in the GEMV.
template <typename T> struct Deq1;                       // one column: (scale bits, zero-point) -> fragment of one word
template <> struct Deq1<f16> {
    f16x2 s2, c1, c2;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        u32x4 o;
        o[0] = f16x2_bits((as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1) * s2);
        o[1] = f16x2_bits((as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2) * s2);
        o[2] = f16x2_bits((as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1) * s2);
        o[3] = f16x2_bits((as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2) * s2);
        return o;
    }
};
template <> struct Deq1<bf16> {
    f16x2 c1, c2;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sbits);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        u32x4 o;
        o[0] = Deq<4, bf16>::scaled_pair(as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1, s);
        o[1] = Deq<4, bf16>::scaled_pair(as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2, s);
        o[2] = Deq<4, bf16>::scaled_pair(as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1, s);
        o[3] = Deq<4, bf16>::scaled_pair(as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2, s);
        return o;
    }
};
template <typename T> struct Mma16;
template <> struct Mma16<f16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(a), as_f16x8(b), c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
    }
};

template <typename T, int RT, int U>
__global__ void __launch_bounds__(1024) gemm_strip16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // W x RT x 1 KiB
    float* red = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int c = lane & 15, kg = lane >> 4;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = blockIdx.z * (16 * RT);
    const int n = strip * 16 + c;                                 // N % 16 == 0 (planner)
    const int S = p.K >> 5;                                       // 32-deep steps in K
    const int b0 = blockIdx.y * p.ksteps_per_split;               // this workgroup's steps [b0, b1)
    const int b1 = min(b0 + p.ksteps_per_split, S);
    const int spw = ((p.ksteps_per_split + W - 1) / W + U - 1) / U * U;
    const int ws = b0 + wave * spw, we = min(ws + spw, b1);

    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qweight, 0, (int)((size_t)p.qrows * p.N * 4), 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, 0, (int)((size_t)(p.K / p.group_size) * p.N * 2), 0x00020000);
    const int zrow_bytes = p.N / 8 * 4;
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.qzeros, 0, (p.K / p.group_size) * zrow_bytes, 0x00020000);
    const unsigned q_lane = ((unsigned)kg * (unsigned)p.N + (unsigned)n) * 4u;
    const unsigned s_lane = (unsigned)n * 2u, z_lane = ((unsigned)n >> 3) * 4u, z_sh = ((unsigned)n & 7u) * 4u;
    const unsigned short* a_src[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        a_src[rt] = (const unsigned short*)p.x + (size_t)min(m0 + rt * 16 + (lane >> 2), p.M - 1) * p.K + (lane & 3) * 8;
    const int a_from = ((c << 2) | kg) << 2;                      // coalesced load layout -> MFMA layout by ds_bpermute (see gemm_stream64_kernel)

    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s0 = ws; s0 < we; s0 += U) {
        unsigned sraw[U], zraw[U], braw[U];
        u32x4 a[U][RT];
#pragma unroll
        for (int j = 0; j < U; ++j) {                             // group constants first: loads return in issue order
            const int sj = min(s0 + j, we - 1);
            const unsigned g = (unsigned)(sj * 32) / (unsigned)p.group_size;
            sraw[j] = __builtin_amdgcn_raw_buffer_load_b16(rsrc_s, s_lane, g * (unsigned)p.N * 2u, 0);
            zraw[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_z, z_lane, g * (unsigned)zrow_bytes, 0);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int sj = min(s0 + j, we - 1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[j][rt] = *(const u32x4*)(a_src[rt] + (size_t)sj * 32);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int sj = min(s0 + j, we - 1);
            braw[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_q, q_lane, (unsigned)sj * 4u * (unsigned)p.N * 4u, 0);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool live = s0 + j < we;
            unsigned z = ((zraw[j] >> z_sh) & 15u) + 1u;
            if (p.zero_mode == GPTQ_ZERO_WRAP) z &= 15u;
            Deq1<T> dq;
            dq.setup(sraw[j] & 0xffffu, z);
            const u32x4 b = dq.frag(braw[j]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                u32x4 t;
#pragma unroll
                for (int d = 0; d < 4; ++d) t[d] = (unsigned)__builtin_amdgcn_ds_bpermute(a_from, (int)a[j][rt][d]);
                u32x4 o;                                          // x in the slot order of the fragments: k0,k4,k1,k5,k2,k6,k3,k7
                o[0] = __builtin_amdgcn_perm(t[2], t[0], 0x05040100u);
                o[1] = __builtin_amdgcn_perm(t[2], t[0], 0x07060302u);
                o[2] = __builtin_amdgcn_perm(t[3], t[1], 0x05040100u);
                o[3] = __builtin_amdgcn_perm(t[3], t[1], 0x07060302u);
                if (!live) o = u32x4{0u, 0u, 0u, 0u};
                acc[rt] = Mma16<T>::run(o, b, acc[rt]);
            }
        }
    }

    // ---- cross-wave sum: C/D layout col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * RT + rt) * 4 + r) * 64 + lane] = acc[rt][r];
    __syncthreads();
    for (int e = tid; e < RT * 256; e += blockDim.x) {             // e = (rt, r, lane')
        const int rt = e >> 8, idx = e & 255, r = idx >> 6, ln = idx & 63;
        float v = 0.f;
        for (int w = 0; w < W; ++w) v += red[(w * RT + rt) * 256 + idx];
        const int m = m0 + rt * 16 + 4 * (ln >> 4) + r;
        const int nn = strip * 16 + (ln & 15);
        if (m >= p.M) continue;
        if (p.ksplit > 1) {
            p.partial[((size_t)blockIdx.y * p.M + m) * p.N + nn] = v;
        } else {
            if (p.bias) v += DType<T>::to_f32(((const T*)p.bias)[nn]);
            ((T*)p.out)[(size_t)m * p.N + nn] = DType<T>::from_f32(v);
        }
    }
}

// ---- batched decode, 4 < M <= 64: 64-column strips, weights by LDS DMA, in-launch K-split combine -----------------------
// The decode stream kernel's structure (gemv.hip: gemv_q4_stream_kernel) on the 16x16x32 matrix core.  One wave DMA
// (global_load_lds_dwordx4, 1 KiB) is exactly one 32-deep K-step of a 64-column strip: 4 packed rows x 256 B, and lane
// (j = l & 15, kg = l >> 4) lands -- and later reads back, lane-linear ds_read_b128, no conflict -- packed row kg of columns
// 4j .. 4j+3.  Word t of those 16 bytes is the lane's B fragment (8 consecutive k of ONE column) of MFMA t, whose "column j"
// is strip column 4j + t: four MFMAs per K-step and 16-row tile, all four fed by the same A fragment.  Nothing about the
// weights touches a VGPR before it is consumed, so a workgroup keeps W x U KiB in flight (16 x 4 = 64 KiB) however many
// accumulators the row tiles need -- which is what the register kernels this replaces ran out of (gemm_strip16_kernel: one dword
// per lane and load; gemv MT = 8: one workgroup per CU): M = 8..64 on 4096 x 11008 ran at 21..27 us for 22.5 MB.
//   A fragments: 16-byte loads of x from L2 per (K-step, row tile), requested before the DMA burst (x is M*K*2 bytes; a
//   64-column strip re-reads it once: N/64 * M * K * 2 bytes of L2 traffic, 22 MB at M = 16 on 4096 x 11008);
//   group constants: 8 B of scales + one zeros word per K-step, Deq1 set up again only when the wave's group changes;
//   waves split the K range of the workgroup; cross-wave sum through LDS in fixed order (slabs alias the landing area);
//   K slices of one tile are combined inside the launch like the decode kernel's: sc1 publish, ticket, last arriver sums
//   the slices in index order.  Tickets: front of the workspace (gptq_mi355x.h), zero before and after every launch.
__device__ __forceinline__ void lds_dma16_nt(const void* gsrc, unsigned lds_dst) {     // see common.cuh: lds_dma16 (this one: nontemporal)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <typename T> __device__ __forceinline__ u32x2 pack4(f32x4 v) {
    struct { T a, b, c, d; } o{DType<T>::from_f32(v[0]), DType<T>::from_f32(v[1]), DType<T>::from_f32(v[2]), DType<T>::from_f32(v[3])};
    return __builtin_bit_cast(u32x2, o);
}

struct S64Seg {
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
struct S64Params {
    S64Seg seg[4];       // up to four layers that read the same x (gptq_forward_multi): the grid runs over all their strips
    const void* x;
    float* partial;      // [ksplit][M][nsum] fp32 when ksplit > 1
    unsigned* tickets;   // one per strip (over all layers)
    int nseg, M, K, group_size, zero_mode, ksplit, ksteps_per_split, nsum;
};

// PIPE (2+ row tiles, M > 16): two landing areas and two register sets per wave -- pass i + 1 (its x fragments, group constants and U KiB of
// weights) is requested BEFORE pass i is computed on.  Without it a wave of the M = 33..64 form alternates "request" and "compute 2 K-steps x
// 16 MFMAs" with nothing in flight while it computes: 8 passes x ~2 us on 4096 x 11008 (the round-2 diagnosis).
template <typename T, int RT, int U, bool PIPE = false>
__global__ void __launch_bounds__(RT == 1 ? 1024 : 512) gemm_stream64_kernel(S64Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, kg = lane >> 4;
    constexpr int STAGES = PIPE ? 2 : 1;
    char* const wq = smem + (size_t)wave * (STAGES * U * 1024);               // this wave's DMA landing area(s)
    const unsigned wq_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(wq));
    // logical block -> (strip over all layers, K slice): slices of one strip are adjacent logical ids (one XCD after the remap)
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = Lb / p.ksplit, ks = Lb - tile * p.ksplit;
    int sI = 0;
    while (sI + 1 < p.nseg && tile >= p.seg[sI].blk_end) ++sI;                // wave-uniform (kernel arguments only)
    const S64Seg& sg = p.seg[sI];
    const int strip = tile - (sI ? p.seg[sI - 1].blk_end : 0);
    const int N = sg.N;
    constexpr int m0 = 0;                                                     // one row tile: M <= 16 * RT (planner)
    const int n0 = strip * 64 + j16 * 4;
    const bool col_ok = n0 < N;
    const int nload = col_ok ? n0 : 0;
    const int S = p.K >> 5;
    const int b0 = ks * p.ksteps_per_split, b1 = min(b0 + p.ksteps_per_split, S);
    const int spw = (b1 - b0 + W - 1) / W;
    const int ws = b0 + wave * spw, we = min(ws + spw, b1);

    const unsigned* __restrict__ qsrc = sg.qweight + (size_t)kg * N + nload;          // + step * 4 * N
    const T* __restrict__ scales = (const T*)sg.scales + nload;
    const unsigned* __restrict__ zsrc = sg.qzeros + (nload >> 3);
    const unsigned z_sh = ((unsigned)nload & 7u) * 4u;                                  // 0 or 16
    const int zrow_words = N >> 3;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;
    const unsigned gsteps = (unsigned)p.group_size >> 5;                                // K-steps per group (group_size % 32 == 0)
    const unsigned short* a_src[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        a_src[rt] = (const unsigned short*)p.x + (size_t)min(m0 + rt * 16 + (lane >> 2), p.M - 1) * p.K + (lane & 3) * 8;
    // A fragments are LOADED with 4 adjacent lanes covering the 64 contiguous bytes (one K-step) of one row -- lane l: row l >> 2,
    // k-octet l & 3: 16 segments of 64 B per instruction -- and moved to the MFMA's layout (lane (i, kg) holds k-octet kg of row i:
    // adjacent lanes = different rows, 64 cache-line lookups per instruction) by 4 ds_bpermute per fragment.  Loading in the MFMA
    // layout directly cost 3-5 us at M = 32 / 64 (timing ablation with coalesced-but-wrong loads: 16.4 -> 13.5, 25.3 -> 20.5 us).
    const int a_from = ((j16 << 2) | kg) << 2;                    // byte address for ds_bpermute: source lane 4 * i + kg

    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    Deq1<T> dq[4];
    int g_cur = -1;

    if constexpr (PIPE) {
        struct Pass { u32x2 sraw[U]; unsigned zw[U]; u32x4 a[U][RT]; int gj[U]; };
        auto issue = [&](int s0, int stage, Pass& P) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < U; ++j) {                         // small L2-resident loads first: they return first
                const int sj = min(s0 + j, we - 1);
                P.gj[j] = (int)((unsigned)sj / gsteps);
                P.sraw[j] = *(const u32x2*)(scales + (size_t)P.gj[j] * N);
                P.zw[j] = zsrc[(size_t)P.gj[j] * zrow_words];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int sj = min(s0 + j, we - 1);
#pragma unroll
#if defined(GPTQ_S64_ABL) && (GPTQ_S64_ABL & 1)
                for (int rt = 0; rt < RT; ++rt) P.a[j][rt] = u32x4{0x3c003c00u + (unsigned)sj, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (unsigned)rt};
#else
                for (int rt = 0; rt < RT; ++rt) P.a[j][rt] = *(const u32x4*)(a_src[rt] + (size_t)sj * 32);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int sj = min(s0 + j, we - 1);
                lds_dma16_nt(qsrc + (size_t)sj * 4 * N, wq_lds + (stage * U + j) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto consume = [&](int s0, int stage, const Pass& P) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const u32x4 qv = *(const u32x4*)(wq + (stage * U + j) * 1024 + lane * 16);
                if (P.gj[j] != g_cur) {                           // wave-uniform: the step index depends on the wave id only
                    g_cur = P.gj[j];
                    const unsigned zz = P.zw[j] >> z_sh;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const unsigned sw = P.sraw[j][t >> 1];
                        dq[t].setup((t & 1) ? (sw >> 16) : (sw & 0xffffu), (((zz >> (4 * t)) & 15u) + 1u) & zmask);
                    }
                }
                const bool live = s0 + j < we;
                u32x4 b[4];
#pragma unroll
#if defined(GPTQ_S64_ABL) && (GPTQ_S64_ABL & 4)
                for (int t = 0; t < 4; ++t) b[t] = u32x4{qv[t], qv[t] ^ 0x11111111u, qv[t] ^ 0x22222222u, qv[t] ^ 0x44444444u};
#else
                for (int t = 0; t < 4; ++t) b[t] = dq[t].frag(qv[t]);
#endif
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    u32x4 x4;
#if defined(GPTQ_S64_ABL) && (GPTQ_S64_ABL & 2)
                    x4 = P.a[j][rt];
                    u32x4 o = x4;
#else
#pragma unroll
                    for (int c = 0; c < 4; ++c) x4[c] = (unsigned)__builtin_amdgcn_ds_bpermute(a_from, (int)P.a[j][rt][c]);
                    u32x4 o;                                      // x in the slot order of the fragments: k0,k4,k1,k5,k2,k6,k3,k7
                    o[0] = __builtin_amdgcn_perm(x4[2], x4[0], 0x05040100u);
                    o[1] = __builtin_amdgcn_perm(x4[2], x4[0], 0x07060302u);
                    o[2] = __builtin_amdgcn_perm(x4[3], x4[1], 0x05040100u);
                    o[3] = __builtin_amdgcn_perm(x4[3], x4[1], 0x07060302u);
#endif
                    if (!live) o = u32x4{0u, 0u, 0u, 0u};
#if defined(GPTQ_S64_ABL) && (GPTQ_S64_ABL & 8)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[rt][t][0] += as_f32((o[t] ^ b[t][t]) & 0x3fffffffu);
#else
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[rt][t] = Mma16<T>::run(o, b[t], acc[rt][t]);
#endif
                }
            }
        };
        // the newer pass leaves 2 U + U RT loads and U DMAs outstanding behind the pass that is about to be computed on
        constexpr int NEWER = 3 * U + U * RT;
        Pass P0, P1;
        if (ws < we) issue(ws, 0, P0);
        for (int s0 = ws; s0 < we; s0 += 2 * U) {
            if (s0 + U < we) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // WAR: the reads of stage 1 two passes ago are done
                issue(s0 + U, 1, P1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            consume(s0, 0, P0);
            if (s0 + U >= we) break;
            if (s0 + 2 * U < we) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue(s0 + 2 * U, 0, P0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            consume(s0 + U, 1, P1);
        }
    } else {
    for (int s0 = ws; s0 < we; s0 += U) {
        u32x2 sraw[U];
        unsigned zw[U];
        u32x4 a[U][RT];
        int gj[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {                             // small L2-resident loads first: they return first
            const int sj = min(s0 + j, we - 1);
            gj[j] = (int)((unsigned)sj / gsteps);
            sraw[j] = *(const u32x2*)(scales + (size_t)gj[j] * N);
            zw[j] = zsrc[(size_t)gj[j] * zrow_words];             // raw word: nothing is computed on loaded values up here
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int sj = min(s0 + j, we - 1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[j][rt] = *(const u32x4*)(a_src[rt] + (size_t)sj * 32);
        }
        if (s0 != ws) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // WAR: last pass's ds_reads are done
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int sj = min(s0 + j, we - 1);
            lds_dma16_nt(qsrc + (size_t)sj * 4 * N, wq_lds + j * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        // step j is consumed as soon as DMA j has landed: vmcnt retires in order and the U DMAs are the wave's youngest VMEM
        // operations, so "at most U-1-j outstanding" means DMAs 0..j and every older load are complete
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] {
                 constexpr int j = J;
                 asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1 - j) : "memory");
                 const u32x4 qv = *(const u32x4*)(wq + j * 1024 + lane * 16);
                 if (gj[j] != g_cur) {                            // wave-uniform: the step index depends on the wave id only
                     g_cur = gj[j];
                     const unsigned zz = zw[j] >> z_sh;
#pragma unroll
                     for (int t = 0; t < 4; ++t) {
                         const unsigned sw = sraw[j][t >> 1];
                         dq[t].setup((t & 1) ? (sw >> 16) : (sw & 0xffffu), (((zz >> (4 * t)) & 15u) + 1u) & zmask);
                     }
                 }
                 const bool live = s0 + j < we;
                 u32x4 b[4];
#pragma unroll
                 for (int t = 0; t < 4; ++t) b[t] = dq[t].frag(qv[t]);
#pragma unroll
                 for (int rt = 0; rt < RT; ++rt) {
                     u32x4 x4;
#pragma unroll
                     for (int c = 0; c < 4; ++c) x4[c] = (unsigned)__builtin_amdgcn_ds_bpermute(a_from, (int)a[j][rt][c]);
                     u32x4 o;                                     // x in the slot order of the fragments: k0,k4,k1,k5,k2,k6,k3,k7
                     o[0] = __builtin_amdgcn_perm(x4[2], x4[0], 0x05040100u);
                     o[1] = __builtin_amdgcn_perm(x4[2], x4[0], 0x07060302u);
                     o[2] = __builtin_amdgcn_perm(x4[3], x4[1], 0x05040100u);
                     o[3] = __builtin_amdgcn_perm(x4[3], x4[1], 0x07060302u);
                     if (!live) o = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                     for (int t = 0; t < 4; ++t) acc[rt][t] = Mma16<T>::run(o, b[t], acc[rt][t]);
                 }
             }()),
             ...);
        }(std::make_integer_sequence<int, U>{});
    }

    }

    // ---- cross-wave sum (LDS slabs over the landing area, fixed order), then write or publish ---------------------------
    // C/D layout of the 16x16 MFMA: column = lane & 15 (-> strip column 4j + t), row = 4 * (lane >> 4) + r.  A lane writes, per
    // (row tile, r), the float4 over t = its 4 adjacent columns of one row: lane-linear 16-byte LDS accesses both ways.
    __syncthreads();                                              // every wave is done with its landing area
    f32x4* const slab = (f32x4*)smem;                             // [W][RT * 4][64]
    constexpr int E = RT * 4 * 64;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            slab[(size_t)wave * E + (rt * 4 + r) * 64 + lane] = f32x4{acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]};
    __syncthreads();
    unsigned* const flag = (unsigned*)(slab + (size_t)W * E);
    const size_t pslab = (size_t)p.M * p.nsum;
    for (int e = tid; e < E; e += blockDim.x) {
        f32x4 v = slab[e];
        for (int w = 1; w < W; ++w) v += slab[(size_t)w * E + e];
        const int ln = e & 63, rr = e >> 6;                       // rr = rt * 4 + r
        const int m = m0 + (rr >> 2) * 16 + 4 * (ln >> 4) + (rr & 3);
        const int n = strip * 64 + (ln & 15) * 4;
        if (m >= p.M || n >= N) continue;
        if (p.ksplit > 1) {
            float* dst = p.partial + (size_t)ks * pslab + (size_t)m * p.nsum + sg.col0 + n;
#pragma unroll
            for (int t = 0; t < 4; ++t) __hip_atomic_store(dst + t, v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
        } else {
            if (sg.bias) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] += DType<T>::to_f32(((const T*)sg.bias)[n + t]);
            }
            *(u32x2*)((T*)sg.out + (size_t)m * N + n) = pack4<T>(v);
        }
    }
    if (p.ksplit > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // every publishing wave drains its stores
        __syncthreads();
        if (tid == 0) *flag = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)(p.ksplit - 1)) return;                        // not the last slice of this tile
        for (int e = tid; e < E; e += blockDim.x) {
            const int ln = e & 63, rr = e >> 6;
            const int m = m0 + (rr >> 2) * 16 + 4 * (ln >> 4) + (rr & 3);
            const int n = strip * 64 + (ln & 15) * 4;
            if (m >= p.M || n >= N) continue;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < p.ksplit; ++k) {                              // fixed order, sc1 loads (bypass this XCD's non-coherent L2 lines)
                const float* src = p.partial + (size_t)k * pslab + (size_t)m * p.nsum + sg.col0 + n;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] += __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (sg.bias) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] += DType<T>::to_f32(((const T*)sg.bias)[n + t]);
            }
            *(u32x2*)((T*)sg.out + (size_t)m * N + n) = pack4<T>(v);
        }
        if (tid == 0) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

// ---- fp32 I/O: exact-f32 matrix core (v_mfma_f32_32x32x2_f32), any bit width ---------------------------------------------
// The reference's Python path is dtype-agnostic (qlinear_cuda_old.py:291-355): an fp32 layer (use_cuda_fp16=False, or the
// act-order natives that force x.float(), qlinear_cuda.py:216-250) dequantises W = scales * (w - z) in fp32 and multiplies in
// fp32.  Same arithmetic here: every weight is the exact fp32 product the reference forms, the products are accumulated by
// the f32-input MFMA (an fmaf chain, bitwise) -- 157 TFLOP/s peak, 1/16 of the fp16 rate, but the weights are read once per
// 128-row tile instead of once per 4 rows of x as in the GEMV this replaces for M > 8.
//   workgroup = 4 waves (2 x 2), tile 128 x 128, wave 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers);
//   x tile [128][32 k] fp32 through LDS (row stride 33 floats: the 32 rows of a ds_read_b32 hit 32 banks), double buffered;
//   a lane owns column (l & 31) of each of its 2 column tiles: it loads that column's packing unit (1 word = 16/8/4 values,
//   3 words = 32 values for 3-bit), extracts the fields, forms s * (w - z) and feeds k-pair (2j + (l >> 5)) to MFMA j.
template <int BITS>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmParams p) {
    constexpr int KPU = Pack<BITS>::vals, UW = Pack<BITS>::words, BK = 32, UPB = BK / KPU, BM = 128, AS = BK + 1;
    __shared__ float As[2][BM * AS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int L = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    const int bm = L % p.nbm, bn = L / p.nbm;
    const int m0 = bm * BM;
    int ncol[2];
    bool col_ok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = bn * 128 + wn * 64 + nt * 32 + l31;
        col_ok[nt] = n < p.N;
        ncol[nt] = col_ok[nt] ? n : 0;
    }
    const float* __restrict__ x = (const float*)p.x;
    const float* __restrict__ scales = (const float*)p.scales;
    const int zrow_words = p.N / 32 * BITS;
    const int ksteps = p.K / BK;
    // x staging: thread t copies 16 consecutive k of row t >> 1 (4 x 16-byte loads; rows past M are clamped, never stored)
    const int a_row = tid >> 1, a_c0 = (tid & 1) * 16;
    const float* a_src = x + (size_t)min(m0 + a_row, p.M - 1) * p.K + a_c0;
    f32x4 a_next[4];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_next[i] = *(const f32x4*)(a_src + (size_t)kt * BK + 4 * i);
    };
    auto store_a = [&](int buf) {
        float* dst = &As[buf][a_row * AS + a_c0];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[4 * i + c] = a_next[i][c];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    load_a(0);
    store_a(0);
    __syncthreads();
    for (int kt = 0; kt < ksteps; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ksteps) load_a(kt + 1);
        // this K-step's packing units of the lane's two columns, and their group constants
        unsigned w[UPB][2][UW];
        float sc[UPB][2];
        int zp[UPB][2];
#pragma unroll
        for (int u = 0; u < UPB; ++u) {
            const int unit = kt * UPB + u;
            const int g = (unit * KPU) / p.group_size;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int i = 0; i < UW; ++i) w[u][nt][i] = p.qweight[(size_t)(unit * UW + i) * p.N + ncol[nt]];
                sc[u][nt] = scales[(size_t)g * p.N + ncol[nt]];
                const int f = (int)stream_field(p.qzeros + (size_t)g * zrow_words, 1, (unsigned)ncol[nt], BITS) + 1;
                zp[u][nt] = (p.zero_mode == GPTQ_ZERO_WRAP) ? (f & (int)Pack<BITS>::maxq) : f;
            }
        }
        const float* arow[2] = {&As[buf][(wm * 64 + l31) * AS + half], &As[buf][(wm * 64 + 32 + l31) * AS + half]};
#pragma unroll
        for (int u = 0; u < UPB; ++u) {
            [&]<int... J>(std::integer_sequence<int, J...>) {       // J = k pair inside the unit: MFMA J consumes k = 2J, 2J + 1
                (([&] {
                     float b[2];
#pragma unroll
                     for (int nt = 0; nt < 2; ++nt) {
                         const int f0 = (int)unit_field<BITS, 2 * J>(w[u][nt]), f1 = (int)unit_field<BITS, 2 * J + 1>(w[u][nt]);
                         b[nt] = sc[u][nt] * (float)((half ? f1 : f0) - zp[u][nt]);      // exact integer difference, one fp32 rounding: the reference's W
                     }
                     float a[2];
#pragma unroll
                     for (int mt = 0; mt < 2; ++mt) a[mt] = arow[mt][u * KPU + 2 * J];
#pragma unroll
                     for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                         for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                 }()),
                 ...);
            }(std::make_integer_sequence<int, KPU / 2>{});
        }
        if (kt + 1 < ksteps) store_a(buf ^ 1);
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if (!col_ok[nt]) continue;
        const float bias = p.bias ? ((const float*)p.bias)[ncol[nt]] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M) ((float*)p.out)[(size_t)m * p.N + ncol[nt]] = acc[mt][nt][r] + bias;
            }
    }
}

// out = sum_s partial[s] (+bias), fixed order; 4 columns per thread
template <typename T>
__global__ void __launch_bounds__(256) gemm_reduce_kernel(const float* __restrict__ partial, const T* __restrict__ bias,
                                                          T* __restrict__ out, int S, int M, int N) {
    const size_t total4 = (size_t)M * N / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < S; ++k) s += *(const f32x4*)(partial + (size_t)k * M * N + i * 4);
        const int nn = (int)((i * 4) % (size_t)N);
        unsigned short o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = s[c];
            if (bias) v += DType<T>::to_f32(bias[nn + c]);
            o[c] = t_bits(DType<T>::from_f32(v));
        }
        u32x2 w = {(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
        *(u32x2*)((unsigned short*)out + i * 4) = w;
    }
}

// x_out[m, i] = x[m, perm[i]] for 2-byte elements: one row per workgroup pass, the row staged in LDS
// (coalesced 16-byte loads and stores; the gather itself runs on the LDS).
// SLOT: each 8-chunk is written in the k-slot order of the 4-bit fp16 B fragments (k0,k4,k1,k5,k2,k6,k3,k7), so the tiled
// GEMM can copy it to LDS verbatim (no v_perm in its K loop).
// x[m, perm[i]] -> out[m, i] for act-order prefill (the role of exllama's column_remap_kernel, exllama/cuda_func/column_remap.cu:9-39).  A 2-byte gather
// from global memory costs the texture path one address per cycle, so the rows are staged in LDS (coalesced 16-byte loads) and gathered there.
// Round 4: R rows per workgroup share ONE read of perm[] (16 KiB for K = 4096 -- twice a row of x: with one row per workgroup the index loads were
// two thirds of the request traffic; 11.7 us = 2.9 TB/s for the 33.5 MB of a 2048 x 4096 x) and the indices stay in registers across the R rows.
template <bool SLOT, int R>
__global__ void __launch_bounds__(256) permute_rows_kernel(const unsigned short* __restrict__ x, const int* __restrict__ perm,
                                                           int M, int K, unsigned short* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* rows = (unsigned short*)smem;              // [R][K]
    const int m0 = blockIdx.x * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = min(m0 + r, M - 1);
        for (int i = threadIdx.x * 8; i < K; i += 256 * 8) *(u32x4*)(rows + (size_t)r * K + i) = *(const u32x4*)(x + (size_t)m * K + i);
    }
    __syncthreads();
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
        const u32x4 p0 = *(const u32x4*)(perm + i), p1 = *(const u32x4*)(perm + i + 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (m0 + r >= M) break;
            const unsigned short* row = rows + (size_t)r * K;
            u32x4 o;
            if constexpr (SLOT) {
                o[0] = (unsigned)row[p0[0]] | ((unsigned)row[p1[0]] << 16);
                o[1] = (unsigned)row[p0[1]] | ((unsigned)row[p1[1]] << 16);
                o[2] = (unsigned)row[p0[2]] | ((unsigned)row[p1[2]] << 16);
                o[3] = (unsigned)row[p0[3]] | ((unsigned)row[p1[3]] << 16);
            } else {
                o[0] = (unsigned)row[p0[0]] | ((unsigned)row[p0[1]] << 16);
                o[1] = (unsigned)row[p0[2]] | ((unsigned)row[p0[3]] << 16);
                o[2] = (unsigned)row[p1[0]] | ((unsigned)row[p1[1]] << 16);
                o[3] = (unsigned)row[p1[2]] | ((unsigned)row[p1[3]] << 16);
            }
            __builtin_nontemporal_store(o, (u32x4*)(out + (size_t)(m0 + r) * K + i));
        }
    }
}

// Round 4, second form: FOUR rows per workgroup held INTERLEAVED in the LDS ([k][4 rows]: 8 bytes per k), so that one ds_read_b64 per index fetches the
// value of all four rows -- 8 LDS reads + 16 v_perm per four 16-byte outputs where the row-major form above takes 32 ds_read_u16 + 32 address adds + 16
// packs (tools/isa: ~88 -> ~32 instructions per four outputs; the kernel is bound by its instruction stream, not by the 33.5 MB it moves).  The load phase
// transposes 4 rows x 8 k in registers (16 v_perm) and writes 64 contiguous bytes.  LDS: 8 K bytes (88 KiB at K = 11008: granted by init_gemm_device).
template <bool SLOT>
__global__ void __launch_bounds__(256) permute_rows4_kernel(const unsigned short* __restrict__ x, const int* __restrict__ perm, int M, int K,
                                                            unsigned short* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];                // [K][4] values
    const int m0 = blockIdx.x * 4;
    const unsigned short* xr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[r] = x + (size_t)min(m0 + r, M - 1) * K;
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
        u32x4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = *(const u32x4*)(xr[r] + i);
        u32x4 o[4];                                                             // 8 k x {rows 0 | 1, rows 2 | 3}
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            o[w][0] = __builtin_amdgcn_perm(v[1][w], v[0][w], 0x05040100u);     // k = 2 w:     row 0 | row 1 << 16
            o[w][1] = __builtin_amdgcn_perm(v[3][w], v[2][w], 0x05040100u);     //              row 2 | row 3 << 16
            o[w][2] = __builtin_amdgcn_perm(v[1][w], v[0][w], 0x07060302u);     // k = 2 w + 1
            o[w][3] = __builtin_amdgcn_perm(v[3][w], v[2][w], 0x07060302u);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) *(u32x4*)(smem + (size_t)i * 8 + w * 16) = o[w];
    }
    __syncthreads();
    for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
        const u32x4 p0 = *(const u32x4*)(perm + i), p1 = *(const u32x4*)(perm + i + 4);
        u32x2 g[8];                                                             // g[j] = the four rows' values at source index j of this piece
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = *(const u32x2*)(smem + (size_t)p0[j] * 8); g[4 + j] = *(const u32x2*)(smem + (size_t)p1[j] * 8); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m0 + r >= M) break;
            const unsigned sel = (r & 1) ? 0x07060302u : 0x05040100u;
            const int h = r >> 1;
            u32x4 o;
            if constexpr (SLOT) {                                                // (k0, k4) (k1, k5) (k2, k6) (k3, k7): the slot order of the B fragments
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = __builtin_amdgcn_perm(g[4 + w][h], g[w][h], sel);
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = __builtin_amdgcn_perm(g[2 * w + 1][h], g[2 * w][h], sel);
            }
            // plain stores: the GEMM behind this pass reads the permuted x at once -- written around the caches (nontemporal, round 4) the 45 MB of a K = 11008
            // call came back from HBM and cost the stream-K prefill kernel 7 us per layer call (profiles/r05_wide_sk_ab_v4.log); the pass itself is no slower
            *(u32x4*)(out + (size_t)(m0 + r) * K + i) = o;
        }
    }
}

template <bool SLOT>
static void launch_permute_rows_r(const unsigned short* x, const int* perm, int M, int K, unsigned short* out, hipStream_t st) {
    if (M >= 1024 && (size_t)K * 8 <= 160 * 1024 && K % 8 == 0) {                // prefill rows: the interleaved four-row form
        hipLaunchKernelGGL((permute_rows4_kernel<SLOT>), dim3((M + 3) / 4), dim3(256), (size_t)K * 8, st, x, perm, M, K, out);
        return;
    }
    // rows per workgroup: as many as keep >= 512 workgroups in the launch and <= 64 KiB of LDS (the default dynamic-LDS limit: no per-function grant needed)
    int r = 1;
    while (r < 4 && (size_t)(2 * r) * K * 2 <= 64 * 1024 && M / (2 * r) >= 512) r *= 2;
    const int blocks = (M + r - 1) / r;
    const size_t lds = (size_t)r * K * 2;
    if (r == 4) hipLaunchKernelGGL((permute_rows_kernel<SLOT, 4>), dim3(blocks), dim3(256), lds, st, x, perm, M, K, out);
    else if (r == 2) hipLaunchKernelGGL((permute_rows_kernel<SLOT, 2>), dim3(blocks), dim3(256), lds, st, x, perm, M, K, out);
    else hipLaunchKernelGGL((permute_rows_kernel<SLOT, 1>), dim3(blocks), dim3(256), lds, st, x, perm, M, K, out);
}

hipError_t launch_permute_rows16(const void* x, const int32_t* perm, int M, int K, void* x_out, hipStream_t st, bool slot_order) {
    if (slot_order) launch_permute_rows_r<true>((const unsigned short*)x, perm, M, K, (unsigned short*)x_out, st);
    else launch_permute_rows_r<false>((const unsigned short*)x, perm, M, K, (unsigned short*)x_out, st);
    return hipGetLastError();
}

// ---- host side ----------------------------------------------------------------------------------
// Balanced tail of the tiled kernel by default?  (tuning.reserved[3] = 40 / 41 forces the rule on / off for A/B runs.)
constexpr bool GEMM_TAIL_DEFAULT = true;

template <int BITS, typename T, int MT, int BK, int VAR, bool XPRE, bool GLDS, int KG>
static constexpr size_t gemm_lds_bytes() { return (size_t)KG * 2 * (32 * MT) * (GLDS ? BK * 2 : BK * 2 + 16) + (VAR == 32 ? 64 : 0); }

template <int BITS, typename T, int MT, int BK, int VAR, bool XPRE, bool GLDS, int KG>
static hipError_t grant_lds() {
    return hipFuncSetAttribute((const void*)gemm_kernel<BITS, T, MT, BK, VAR, XPRE, GLDS, KG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)gemm_lds_bytes<BITS, T, MT, BK, VAR, XPRE, GLDS, KG>());
}

// Per-device, once, outside any capture (gptq_init): kernels whose dynamic LDS exceeds the 64 KiB default.
template <typename T, int RT, int U> static hipError_t grant_stream64() {
    return hipFuncSetAttribute((const void*)gemm_stream64_kernel<T, RT, U, (RT > 1)>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T> static hipError_t grant_stream64_t() {
    hipError_t e = hipSuccess;
    auto acc = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    acc(grant_stream64<T, 1, 2>()); acc(grant_stream64<T, 1, 4>()); acc(grant_stream64<T, 1, 8>());
    acc(grant_stream64<T, 2, 2>()); acc(grant_stream64<T, 2, 4>());
    acc(grant_stream64<T, 4, 1>());
    return e;
}

hipError_t init_gemm_device() {
    hipError_t e = grant_lds<4, f16, 4, 64, 1, true, true, 2>();
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)permute_rows4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)permute_rows4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = grant_lds<4, f16, 4, 64, 1, false, false, 2>();
    if (e == hipSuccess) e = grant_lds<4, bf16, 4, 64, 1, false, false, 2>();
#ifdef GPTQ_GEMM_ABLATIONS
    if (e == hipSuccess) e = grant_lds<4, f16, 4, 64, 16, false, false, 2>();
    if (e == hipSuccess) e = grant_lds<4, f16, 4, 64, 32, false, false, 2>();
    if (e == hipSuccess) e = grant_lds<4, f16, 4, 64, 32, true, true, 2>();
#endif
    if (e == hipSuccess) e = grant_stream64_t<f16>();
    if (e == hipSuccess) e = grant_stream64_t<bf16>();
    return e;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- batched decode planner (gemm_stream64_kernel), shared by the single-layer path (plan_gemm) and gptq_forward_multi ------
// Measured (tools/stream64_sweep.py, profiles/r02_stream64_sweep*.log; us per launch, M = 8 / 16 / 32 / 64, best older kernel -> this one):
//   4096x11008  18.2 / 19.0 / 23.4 / 26.5 -> 10.3 / 11.2 / 15.1 / 23.4   (172 strips, no K split)
//   11008x4096  16.7 / 20.2 / 22.6 / 27.6 -> 11.7 / 12.1 / 16.7 / 25.7   (64 strips x 4 K slices)
//   5120x5120   15.7 / 15.9 / 19.0 / 23.1 ->  9.8 / 10.3 / 13.8 / 19.9   (80 x 3)      8192x3584  13.2 / 15.2 / 17.3 / 21.7 -> 10.4 / 11.1 / 14.1 / 21.2 (56 x 4)
//   3584x8192   12.6 / 12.8 / 14.0 / 18.7 ->  9.4 /  9.8 / 13.4 / 19.5   (128 x 2)     8192x28672 51.7 / 54.3 / 65.3 / 72.2 -> 33.5 / 36.5 / 50.2 / 81.0 (448 x 1)
//   4096x4096    7.9 /  8.9 / 12.2 / 15.7 ->  8.7 /  9.2 / 12.1 / 16.7   (64 x 4: the 16-column / skinny kernels stay)
//   8192x1024    8.9 /  9.3 / 10.8 / 14.2 ->  9.5 / 10.1 / 12.2 / 17.9   (16 x 4..8: too few workgroups; older kernels stay)
// K slices: as many as keep strips x slices <= 256 (one 16-wave workgroup per CU: a 257th starts a second round; 8192x3584 with
// 5 slices = 280 workgroups: 16.0 us, with 4 = 224: 10.4), none from 160 strips up (the combine costs more than the idle CUs).
Stream64Plan plan_stream64(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* tune) {
    Stream64Plan pl{};
    if (n < 1 || n > 4 || M < 1 || M > 64) return pl;
    const gptq_layer_t& A = *Ls[0];
    int strips = 0, nsum = 0, nmax = 0;
    for (int i = 0; i < n; ++i) {
        const gptq_layer_t& L = *Ls[i];
        if (L.bits != 4 || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16) || L.epilogue != GPTQ_EPI_NONE) return pl;
        if (L.K % 32 || L.N % 32 || L.group_size % 32) return pl;
        if (L.g_idx != nullptr && (n > 1 || !L.qweight_seq || !L.perm)) return pl;      // act-order: single layers only (x is permuted per layer)
        if (L.K != A.K || L.group_size != A.group_size || L.dtype != A.dtype || L.zero_mode != A.zero_mode) return pl;
        strips += (L.N + 63) / 64;
        nsum += L.N;
        nmax = L.N > nmax ? L.N : nmax;
    }
    if ((size_t)strips * 4 > WS_HEADER_EPOCH_OFFSET) return pl;                              // one ticket per strip
    pl.nseg = n;
    pl.strips_total = strips;
    pl.nsum = nsum;
    pl.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);                                          // row tiles of 16 in the one row tile of the launch
    const int S = A.K / 32;
    pl.ksteps_total = S;
    int ks = (tune && tune->ksplit > 0 && tune->path == 3) ? tune->ksplit : 0;
    if (!ks) {
        ks = strips >= 160 ? 1 : 256 / strips;
        if (ks > 8) ks = 8;
        while (ks > 1 && S / ks < 8) --ks;                                            // at least 8 K-steps per slice
    }
    if (ks > S) ks = S;
    if (ks < 1) ks = 1;
    pl.ksteps_per_split = (S + ks - 1) / ks;
    pl.ksplit = (S + pl.ksteps_per_split - 1) / pl.ksteps_per_split;                  // no empty slices
    // one row tile of 16: 16 waves while the launch is one round of one workgroup per CU, 8 (two or three workgroups per CU) beyond;
    // 2+ row tiles: > 128 registers per lane (__launch_bounds__(512)), cross-wave slabs W x RT x 4 KiB <= 128 KiB
    int waves = (tune && tune->waves > 0) ? tune->waves : ((pl.mt == 1 && (long)strips * pl.ksplit <= 256) ? 16 : 8);
    if (pl.mt > 1 && waves > 8) waves = 8;
    if (waves > 16) waves = 16;
    pl.waves = waves;
    int u = (tune && tune->reserved[0] > 0) ? tune->reserved[0] : (pl.mt == 4 ? 1 : 2);   // 4 row tiles: two pipelined passes of ONE K-step each (U = 2 spills 44 registers)
    if (pl.mt == 1) u = u >= 8 ? 8 : (u >= 4 ? 4 : 2);
    else if (pl.mt == 2) u = u >= 4 ? 4 : 2;
    else u = 1;                                   // (the 2- / 4-step forms of 4 row tiles spilled 44..150 registers: retired)
    pl.u = u;
    const size_t land = (size_t)waves * u * 1024 * (pl.mt > 1 ? 2 : 1), slabs = (size_t)waves * pl.mt * 4096;     // 2+ row tiles: pipelined passes, two landing areas
    pl.lds_bytes = (land > slabs ? land : slabs) + 16;
    if (pl.lds_bytes > 160 * 1024) return pl;
    pl.partial_bytes = pl.ksplit > 1 ? (size_t)pl.ksplit * M * nsum * sizeof(float) : 0;
    const bool small = n == 1 && A.N <= 4096 && A.K <= 4096;      // <= 8.8 MB: one round of 16-column strips / the skinny kernel is as fast
    pl.pays = (long)strips * pl.ksplit >= 160 && !small && (M <= 32 || nmax < 12288);   // 33+ rows on very wide layers: the tiled kernel
    pl.ok = true;
    return pl;
}

template <typename T, int RT, int U>
static hipError_t launch_stream64_one(const Stream64Plan& pl, const S64Params& p, hipStream_t st) {
    // > 64 KiB of LDS for most shapes: granted by init_gemm_device() (gptq_init)
    hipLaunchKernelGGL((gemm_stream64_kernel<T, RT, U, (RT > 1)>), dim3(pl.strips_total * pl.ksplit), dim3(pl.waves * 64), pl.lds_bytes, st, p);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_stream64_t(const Stream64Plan& pl, const S64Params& p, hipStream_t st) {
    switch (pl.mt * 16 + pl.u) {
        case 16 + 2: return launch_stream64_one<T, 1, 2>(pl, p, st);
        case 16 + 4: return launch_stream64_one<T, 1, 4>(pl, p, st);
        case 16 + 8: return launch_stream64_one<T, 1, 8>(pl, p, st);
        case 32 + 2: return launch_stream64_one<T, 2, 2>(pl, p, st);
        case 32 + 4: return launch_stream64_one<T, 2, 4>(pl, p, st);
        case 64 + 1: return launch_stream64_one<T, 4, 1>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_stream64(const gptq_layer_t* const* Ls, const Stream64Plan& pl, const void* x, void* const* outs, int M,
                           void* ws_header, void* partial, const uint32_t* qweight_override, hipStream_t st) {
    if (!pl.ok) return hipErrorNotSupported;
    S64Params p{};
    int blk = 0, col = 0;
    for (int i = 0; i < pl.nseg; ++i) {
        const gptq_layer_t& L = *Ls[i];
        S64Seg& sg = p.seg[i];
        sg.qweight = (i == 0 && qweight_override) ? qweight_override : L.qweight;
        sg.qzeros = L.qzeros;
        sg.scales = L.scales;
        sg.bias = L.bias;
        sg.out = outs[i];
        sg.N = L.N;
        blk += (L.N + 63) / 64;
        sg.blk_end = blk;
        sg.col0 = col;
        col += L.N;
    }
    p.x = x;
    p.partial = (float*)partial;
    p.tickets = (unsigned*)ws_header;
    p.nseg = pl.nseg; p.M = M; p.K = Ls[0]->K; p.group_size = Ls[0]->group_size; p.zero_mode = Ls[0]->zero_mode;
    p.ksplit = pl.ksplit; p.ksteps_per_split = pl.ksteps_per_split; p.nsum = pl.nsum;
    return (Ls[0]->dtype == GPTQ_F16) ? launch_stream64_t<f16>(pl, p, st) : launch_stream64_t<bf16>(pl, p, st);
}

GemmPlan plan_gemm(const gptq_layer_t& L, int M, const gptq_tuning_t* tune) {
    GemmPlan pl{};
    const int kpu = unit_vals(L.bits);
    const bool seq = (L.g_idx == nullptr) || (L.qweight_seq != nullptr && L.perm != nullptr);
    if (L.dtype == GPTQ_F32) {               // exact-f32 matrix core, 128 x 128 tiles, any bit width; groups made of whole packing units
        pl.f32 = true;
        pl.supported = seq && (L.group_size % kpu == 0) && (L.K % 32 == 0) && (L.N % 32 == 0);
        if (!pl.supported) return pl;
        pl.use_seq = (L.g_idx != nullptr);
        pl.xperm_bytes = pl.use_seq ? align_up((size_t)M * L.K * 4, 256) : 0;
        pl.mt = 4; pl.bk = 32; pl.bm = 128; pl.bn = 128; pl.waves = 4; pl.kg = 1;
        pl.nbm = (M + 127) / 128;
        pl.nbn = (L.N + 127) / 128;
        pl.ksplit = 1;
        pl.ksteps_total = pl.ksteps_per_split = L.K / 32;
        pl.workspace_bytes = pl.xperm_bytes;
        return pl;
    }
    pl.supported = (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16) && seq && (L.group_size % 32 == 0) && (L.K % 32 == 0) &&
                   (L.N % 32 == 0) && (L.group_size % kpu == 0) && ((size_t)L.K * 2 <= 64 * 1024 || L.g_idx == nullptr);
    if (!pl.supported) return pl;
    pl.use_seq = (L.g_idx != nullptr);
    pl.xperm_bytes = pl.use_seq ? align_up((size_t)M * L.K * 2, 256) : 0;
    const int force_skinny = tune ? tune->reserved[2] : 0;      // experiment knob: 1 = skinny, 2 = tiled
    // 64 .. ~1024 rows of a layer that carries its decode copy where its tiles fill the chip (panel_pays): whole-K panels of 64 x 32 nt tiles, no exchange
    // (gemm_panel.hip); lab knobs 52 / 53 force it on / off.  Asked before the rows kernel: from ~96 rows it is the faster of the two (profiles/r06_panel_sweep.log)
    {
        const int knob = tune ? tune->reserved[3] : 0;
        if (knob == GPTQ_LAB_VARIANT_PANEL_ON || (knob != GPTQ_LAB_VARIANT_PANEL_OFF && (!tune || tune->path != 3 || knob == 0) && force_skinny == 0 && panel_pays(L, M))) {
            const PanelPlan pp = plan_panel(L, M, knob == GPTQ_LAB_VARIANT_PANEL_ON ? tune : nullptr);
            if (pp.ok) {
                pl.panel = true;
                pl.panelp = pp;
                pl.xnat = pl.use_seq;
                pl.mt = pp.mt; pl.bk = 64; pl.bm = 32 * pp.mt; pl.bn = 32 * pp.nt; pl.nbm = pp.nbm; pl.nbn = pp.nbn;
                pl.waves = pp.kp; pl.u = 2; pl.kg = pp.kp;
                pl.ksplit = 1; pl.ksteps_total = pl.ksteps_per_split = L.K / 64;
                pl.workspace_bytes = pl.xperm_bytes;
                return pl;
            }
        }
    }
    // 5 .. ~256 rows of a layer that carries its decode copy: whole-K workgroups, no exchange (gemm_rows.hip); lab knobs 50 / 51 force it on / off
    {
        const int knob = tune ? tune->reserved[3] : 0;
        if (knob == GPTQ_LAB_VARIANT_ROWS_ON || (knob != GPTQ_LAB_VARIANT_ROWS_OFF && (!tune || tune->path != 3 || knob == 0) && force_skinny == 0 && rows_pays(L, M))) {
            const RowsPlan rp = plan_rows(L, M, knob == GPTQ_LAB_VARIANT_ROWS_ON ? tune : nullptr);
            if (rp.ok) {
                pl.rows = true;
                pl.rowsp = rp;
                pl.xnat = pl.use_seq;
                pl.mt = rp.rb; pl.bk = 128; pl.bm = 16 * rp.rb; pl.bn = 16 * rp.s; pl.nbm = rp.npm; pl.nbn = rp.nsg;
                pl.waves = rp.waves; pl.u = 2; pl.kg = 1;
                pl.ksplit = 1; pl.ksteps_total = pl.ksteps_per_split = L.K / 128;
                pl.workspace_bytes = pl.xperm_bytes;
                return pl;
            }
        }
    }
    // Measured crossovers (tools/midm_bench.py, us per launch at M = 9/16/32/64):
    //   4096x4096   strips16  8.0/ 9.1/12.3/19.6   skinny64 12.1/12.3/12.7/15.5   tiled 15.5/15.7/17.1/22.6
    //   11008x4096  strips16 17.9/20.1/27.9/47.9   skinny64 23.2/23.1/24.0/27.4   tiled 27.6/28.0/28.8/37.2
    //   4096x11008  strips16 19.5/22.4/31.4/53.7   skinny64 19.2/19.5/22.8/33.0   tiled 17.8/18.7/20.9/25.7
    // 16-column strips re-read x from L2 once per strip (M*K*N/8 bytes), so they only pay up to M = 16; a wide N gives the
    // tiled kernel enough 256-column tiles to fill the chip without help.
    pl.skinny = (force_skinny == 1) || (force_skinny == 0 && M <= 64 && (L.N + 255) / 256 < 32);
    if (pl.skinny && M > 128) pl.skinny = false;
    // 17 .. 128 rows: everything by LDS DMA (gemm_mid.hip)
    {
        const gptq_layer_t* one[1] = {&L};
        const MidPlan mp = plan_mid(one, 1, M, tune);
        // by default only without kernel-specific experiment knobs: tuning.reserved[0..3] mean different things to the tiled / stream64 kernels
        const bool plain_tuning = !tune || (tune->reserved[0] == 0 && tune->reserved[1] == 0 && tune->reserved[3] == 0);
        pl.mid = mp.ok && (force_skinny == 5 || (force_skinny == 0 && mp.pays && plain_tuning));
        if (pl.mid) {
            pl.skinny = false;
            pl.midp = mp;
            pl.mt = mp.rt; pl.bk = 32; pl.bm = 16 * mp.rt; pl.bn = 64 * mp.cw;
            pl.nbm = mp.row_blocks; pl.nbn = mp.strips_total;
            pl.waves = mp.waves; pl.u = mp.stages;
            pl.ksteps_total = mp.ksteps_total; pl.ksteps_per_split = mp.ksteps_per_split; pl.ksplit = mp.ksplit;
            pl.workspace_bytes = pl.xperm_bytes + mp.partial_bytes;
            return pl;
        }
    }
    // batched decode (4 < M <= 64, 4-bit): 64-column strips, weights by LDS DMA, K slices combined inside the launch
    // batched decode (4 < M <= 64, 4-bit): 64-column strips, weights by LDS DMA, K slices combined inside the launch
    {
        const gptq_layer_t* one[1] = {&L};
        const Stream64Plan sp = plan_stream64(one, 1, M, tune);
        pl.stream64 = sp.ok && (force_skinny == 4 || (force_skinny == 0 && M >= 3 && sp.pays));      // 3..4 rows: want_gemm takes it only unsplit
        if (pl.stream64) {
            pl.skinny = false;
            pl.mt = sp.mt; pl.bk = 32; pl.bm = 16 * sp.mt; pl.bn = 64;
            pl.nbm = 1; pl.nbn = sp.strips_total;
            pl.waves = sp.waves; pl.u = sp.u;
            pl.ksteps_total = sp.ksteps_total; pl.ksteps_per_split = sp.ksteps_per_split; pl.ksplit = sp.ksplit;
            pl.workspace_bytes = pl.xperm_bytes + sp.partial_bytes;
            return pl;
        }
    }
    // 16-column strips on the 16x16x32 matrix core: 4-bit fp16/bf16, M <= 64
    const bool strip16_ok = L.bits == 4 && M <= 64 && L.group_size % 32 == 0 && L.K % 32 == 0 && L.N % 16 == 0;
    pl.strip16 = strip16_ok && (force_skinny == 3 || (force_skinny == 0 && M <= 16 && L.N <= 8192));
    if (pl.strip16) {
        pl.skinny = false;
        pl.mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);                  // row tiles of 16
        pl.bk = 32;
        pl.bm = 16 * pl.mt;
        pl.bn = 16;
        pl.nbm = (M + pl.bm - 1) / pl.bm;
        pl.nbn = L.N / 16;
        pl.waves = 16;
        const int S = L.K / 32;
        int ks = (tune && tune->ksplit > 0 && tune->path == 3) ? tune->ksplit : 0;
        if (!ks) {
            ks = 1;
            while ((long)pl.nbm * pl.nbn * ks < 192 && S / (ks * 2) >= pl.waves) ks *= 2;
        }
        if (ks > S) ks = S;
        pl.ksteps_total = S;
        pl.ksteps_per_split = (S + ks - 1) / ks;
        pl.ksplit = (S + pl.ksteps_per_split - 1) / pl.ksteps_per_split;
        pl.workspace_bytes = pl.xperm_bytes + (pl.ksplit > 1 ? (size_t)pl.ksplit * M * L.N * sizeof(float) : 0);
        return pl;
    }
    if (pl.skinny) {
        pl.mt = M <= 32 ? 1 : (M <= 64 ? 2 : 4);
        pl.bk = (L.group_size % 64 == 0 && L.K % 64 == 0) ? 64 : 32;      // = 16 * UNR
        pl.bm = 32 * pl.mt;
        pl.bn = 64;
        pl.nbm = (M + pl.bm - 1) / pl.bm;
        pl.nbn = (L.N + 63) / 64;
        pl.waves = 8;
        const int S = L.K / 16;
        int ks = (tune && tune->ksplit > 0 && tune->path == 3) ? tune->ksplit : 0;
        if (!ks) {
            ks = 1;
            while ((long)pl.nbm * pl.nbn * ks < 224 && S / (ks * 2) >= pl.waves * 4) ks *= 2;
        }
        if (ks > S) ks = S;
        pl.ksteps_total = S;
        pl.ksteps_per_split = (S + ks - 1) / ks;
        pl.ksteps_per_split = (pl.ksteps_per_split + pl.bk / 16 - 1) / (pl.bk / 16) * (pl.bk / 16);   // chunk aligned
        pl.ksplit = (S + pl.ksteps_per_split - 1) / pl.ksteps_per_split;
        pl.workspace_bytes = pl.xperm_bytes + (pl.ksplit > 1 ? (size_t)pl.ksplit * M * L.N * sizeof(float) : 0);
        return pl;
    }
    pl.mt = M <= 32 ? 1 : (M <= 64 ? 2 : 4);
    pl.bk = (L.bits == 4 && pl.mt == 4 && L.K % 64 == 0 && L.group_size % 64 == 0) ? 64 : 32;
    if (tune && tune->reserved[1] == 32) pl.bk = 32;
    pl.variant = tune ? tune->reserved[3] : 0;            // experiment knob: inner-loop schedule variant
    // lab switches (include/gptq_mi355x_lab.h): the 128 x 512 kernel off / forced; by the rule / forced on the checkpoint rows even when the layer has a decode copy
    const int wide_knob = (pl.variant >= GPTQ_LAB_VARIANT_WIDE_OFF && pl.variant <= GPTQ_LAB_VARIANT_WIDE_ROWS_ON) ? pl.variant : 0;
    if (wide_knob) pl.variant = 0;
    const int tail_knob = (pl.variant >= GPTQ_LAB_VARIANT_TAIL_ON && pl.variant <= GPTQ_LAB_VARIANT_TAIL_NO_LIMIT) ? pl.variant : 0;      // balanced tail: by the rule below / off / the rule without its tile limit
    if (tail_knob) pl.variant = 0;
    const int sk_knob = (pl.variant == GPTQ_LAB_VARIANT_WIDE_SK_ON || pl.variant == GPTQ_LAB_VARIANT_WIDE_SK_OFF) ? pl.variant : 0;
    if (sk_knob) pl.variant = 0;
    pl.bm = 32 * pl.mt;
    pl.bn = 256;
    pl.nbm = (M + pl.bm - 1) / pl.bm;
    pl.nbn = (L.N + 255) / 256;
    pl.ksteps_total = L.K / pl.bk;
    int ks = (tune && tune->ksplit > 0 && tune->path == 3) ? tune->ksplit : 0;
    if (!ks) {
        ks = 1;
        const long blocks = (long)pl.nbm * pl.nbn;
        // stop doubling once the next step would pass 256 workgroups: up to 256 the workgroup can run two K groups (8 waves), which
        // beats twice the slices (tools/ksplit_sweep.py, us per layer: 4096x11008 M = 128: 4 slices x 2 groups 30.5 vs 8 slices 35.4;
        // M = 256: 46.4 vs 50.3; M = 512: 1 x 2 72.6 vs 2 slices 79.0)
        while (blocks * ks < 192 && ks < 8 && pl.ksteps_total / (ks * 2) >= 4 && blocks * ks * 2 <= 256) ks *= 2;
    }
    if (ks > pl.ksteps_total) ks = pl.ksteps_total;
    pl.ksteps_per_split = (pl.ksteps_total + ks - 1) / ks;
    pl.ksplit = (pl.ksteps_total + pl.ksteps_per_split - 1) / pl.ksteps_per_split;   // no empty slices
    pl.workspace_bytes = pl.xperm_bytes + (pl.ksplit > 1 ? (size_t)pl.ksplit * M * L.N * sizeof(float) : 0);
    // act-order + the 4-bit fp16 128x256x64 kernel: the permute pre-pass delivers x in k-slot order
    pl.xslot = pl.use_seq && L.bits == 4 && L.dtype == GPTQ_F16 && pl.mt == 4 && pl.bk == 64 && (pl.variant == 0 || pl.variant == 3 || pl.variant == 5 || pl.variant == 6 || pl.variant == 7 || pl.variant == 32);
    pl.glds = pl.xslot && pl.variant != 5;            // variant 5 (experiment): register-staged x
    // At most one tile per CU: run the tile's K range as two concurrent halves inside the workgroup (8 waves).
    const bool even_slices = pl.ksteps_total % pl.ksteps_per_split == 0 && pl.ksteps_per_split % 2 == 0 && pl.ksteps_per_split >= 4;
    const bool kg_ok = L.bits == 4 && pl.bk == 64 && pl.mt == 4 && even_slices && (pl.variant == 0 || pl.variant == 6 || pl.variant == 7 || pl.variant == 16 || pl.variant == 17 || pl.variant == 32) &&
                       (!pl.use_seq || pl.xslot == pl.glds);
    // Balanced tail (see gemm_kernel, TAIL): the tiles past the last full round of 256 are cut into 2 / 4 / 8 K slices when the time model says
    // it pays -- a round costs ksteps x ~1.05 us, the tail round shrinks to ceil(tail * s / 256) / s of it, and the fix-up moves 128 KiB per
    // published slice and direction at ~3 TB/s plus ~1 us per slice on the last arrival.  Measured
    // (profiles/r03_gemm_balanced_tail_ab.log, us per layer, whole tiles -> balanced): 4096x4096 M = 2176 / 2560 / 2944 / 4224: 111 -> 87,
    // 124 -> 99, 122 -> 106, 182 -> 153; 4096x11008 M = 768 / 1024 / 1536 / 2304: 106 -> 85, 120 -> 102, 187 -> 161, 242 -> 221;
    // 11008x4096 M = 2176: 281 -> 221.  A tail the model refuses (4096x11008 at M = 2048: 176 tiles) measured 0.97 - 1.0x with two slices.
    pl.tail = 0; pl.tail_lg = 0;
    {
        const long tiles = (long)pl.nbm * pl.nbn, rem = tiles % 256;
        // every 128-row form of the kernel: 4-bit BK = 64 (act-order fp16 only in its DMA-staged form) and the BK = 32 forms of 2- / 3- / 8-bit and g32 layers
        const bool legal = pl.mt == 4 && pl.ksplit == 1 && pl.variant == 0 && L.N % 256 == 0 &&
                           (pl.bk == 32 || !pl.use_seq || (pl.xslot && pl.glds) || L.dtype == GPTQ_BF16);
        const bool wanted = tail_knob == GPTQ_LAB_VARIANT_TAIL_ON || tail_knob == GPTQ_LAB_VARIANT_TAIL_NO_LIMIT || (tail_knob == 0 && GEMM_TAIL_DEFAULT);
        if (legal && wanted && tiles > 256 && (tiles <= 1024 || tail_knob == GPTQ_LAB_VARIANT_TAIL_NO_LIMIT) && rem > 0) {      // above ~4 rounds: +5 - 7 % in one harness, -4 % in another
            const double t_round = 1.05 * (L.K / 64.0) * (L.bits != 4 ? 1.35 : (pl.bk == 32 ? 1.1 : 1.0));      // 2- / 3- / 8-bit prefill: 660 - 770 TFLOP/s
            double best = 0.0;
            for (int lg = 1; lg <= 3; ++lg) {
                const int s = 1 << lg;
                if (pl.ksteps_total % s != 0 || (pl.ksteps_total / s) * pl.bk < 256) break;
                const double gain = t_round * (1.0 - (double)((rem * s + 255) / 256) / s);
                const double cost = (double)rem * (s - 1) * (2.0 * 131072.0) / 3.0e6 + 1.0 * (s - 1);      // us
                if (gain - cost > best + 0.5) { best = gain - cost; pl.tail_lg = lg; }
            }
            // The model is about twice too optimistic on the gain (two co-resident workgroups overlap: a round of two is shorter than two rounds); the
            // measured gains are 0.35 - 0.75 of the predicted ones at every round count.  Accept from 7.5 % predicted = ~3 % real: 4096x11008 at
            // M = 4096 (96 tiles x 2, 6.7 % predicted) measured +5 % in one session and -5 % in the next and stays on whole tiles.
            if (best >= 0.075 * t_round * (double)tiles / 256.0) pl.tail = (int)rem;
            else pl.tail_lg = 0;
        }
        if (pl.tail) pl.workspace_bytes = pl.xperm_bytes + ((size_t)pl.tail << pl.tail_lg) * (32 * 256 * 16);
    }
    // 128 x 512 tiles with a 128 x 128 tile per wave (gemm_wide.hip) where they fill the chip: whole rounds of 256, or enough rounds that the last one
    // hardly matters (tools/widelab, us per layer, 128 x 256 -> 128 x 512: M = 4096 on 4096^2 131.5 -> 122.8 (256 tiles), 11008x4096 337 -> 314 (256),
    // 4096x11008 360 -> 342 (704); M = 2048 on 4096x11008 (352 tiles = 1.4 rounds) 187 -> 214 and on 4096^2 (128 tiles) 67 -> 102: those stay here)
    {
        const long wt = (long)((M + 127) / 128) * ((L.N + 511) / 512);
        const long rounds = (wt + 255) / 256;
        const bool fills = wt >= 256 && (double)wt / (double)(rounds * 256) >= 0.9;
        // layers that carry their decode copy: weights from it, raw x by LDS DMA (gemm_wide_kernel<T, true, true>); knob 46 / 47 keep the checkpoint rows (A/B).
        // Act-order layers: the copy is made of the re-sequenced rows, so x is permuted in NATURAL order (xnat) instead of the row form's slot order -- and
        // bf16 act-order layers, which have no slot-ordered permute, get the wide tiles this way.
        const bool copy_ok = L.qweight_tiled != nullptr && L.tiled_cols == GPTQ_STRIP_COLS && L.N % GPTQ_STRIP_COLS == 0 && wide_knob != GPTQ_LAB_VARIANT_WIDE_ROWS && wide_knob != GPTQ_LAB_VARIANT_WIDE_ROWS_ON;
        pl.wide = wide_knob != GPTQ_LAB_VARIANT_WIDE_OFF && (fills || wide_knob == GPTQ_LAB_VARIANT_WIDE_ON || wide_knob == GPTQ_LAB_VARIANT_WIDE_ROWS_ON) && pl.mt == 4 && pl.bk == 64 && pl.ksplit == 1 && pl.variant == 0 && tail_knob == 0 &&
                  (wide_gemm_ok(L, M, pl.use_seq, pl.xslot && pl.glds) || (copy_ok && wide_gemm_ok(L, M, false, false)));
        // The same wave tile as a stream-K partition of 128 x 256 tiles (gemm_wide_sk.hip) everywhere else from ~768 rows: every CU runs the same number of
        // 128-deep K-chunks whatever the tile count -- BASELINE config 3 (M = 2048: 256 / 688 / 256 tiles on the three Llama-7B shapes).  Decode-copy layers
        // only (act-order: x permuted in natural order).
        if (pl.mt == 4 && pl.variant == 0 && wide_knob == 0 && tail_knob == 0 && sk_knob != GPTQ_LAB_VARIANT_WIDE_SK_OFF && copy_ok &&      // (wide_sk_ok has the kernel's own conditions: 3 / 4 bits, 32-wide groups too)
            !(tune && tune->ksplit > 0 && tune->path == 3 && tune->ksplit != 1) && wide_sk_ok(L, M) &&
            (sk_knob == GPTQ_LAB_VARIANT_WIDE_SK_ON || (!pl.wide && wide_sk_pays(L, M)))) {
            pl.wsk = true;
            pl.wskg = wide_sk_geom(L, M);
            pl.wide = false; pl.wide_tiled = true; pl.xnat = pl.use_seq; pl.xslot = false; pl.glds = true;
            pl.bk = 64; pl.ksteps_total = L.K / 64;
            pl.tail = 0; pl.tail_lg = 0; pl.ksplit = 1; pl.ksteps_per_split = pl.ksteps_total; pl.kg = 2;
            pl.bn = 256; pl.nbm = pl.wskg.nbm; pl.nbn = pl.wskg.nbn;
            pl.workspace_bytes = pl.xperm_bytes + pl.wskg.slot_bytes;
            return pl;
        }
        if (pl.wide) {
            pl.wide_tiled = copy_ok;
            pl.xnat = pl.wide_tiled && pl.use_seq;
            pl.tail = 0; pl.tail_lg = 0;
            pl.bn = 512; pl.nbn = (L.N + 511) / 512;
            pl.workspace_bytes = pl.xperm_bytes;
            pl.kg = 1;
            return pl;
        }
    }
    pl.kg = (kg_ok && pl.tail == 0 && pl.variant != 6 && pl.variant != 16 && ((long)pl.nbm * pl.nbn * pl.ksplit <= 256 || pl.variant == 7 || pl.variant == 17 || pl.variant == 32)) ? 2 : 1;   // 16 / 17: gemmlab timeline variants (one / two K groups); 32: ping-pong
    return pl;
}

template <int BITS, typename T, int MT, int BK, int VAR = 1, bool XPRE = false, bool GLDS = false, int KG = 1>
static hipError_t launch_one(const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    const size_t lds = gemm_lds_bytes<BITS, T, MT, BK, VAR, XPRE, GLDS, KG>();       // KG = 2: >= the 64 KiB exchange area
    if constexpr (MT == 4 && VAR == 1 && KG == 1) {
        if (pl.tail > 0) {                         // whole tiles first, then the K slices of the tail tiles
            hipLaunchKernelGGL((gemm_kernel<BITS, T, MT, BK, VAR, XPRE, GLDS, 1, true>), dim3(pl.nbm * pl.nbn - pl.tail + (pl.tail << pl.tail_lg), 1),
                               dim3(256), lds, st, p);
            return hipGetLastError();
        }
    }
    auto* kern = gemm_kernel<BITS, T, MT, BK, VAR, XPRE, GLDS, KG>;
    // KG = 2 asks for > 64 KiB of dynamic LDS: granted per function and device by init_gemm_device() (gptq_init), never here --
    // the launch path makes no runtime-API call besides the launch itself, so it is legal under stream capture.
    hipLaunchKernelGGL(kern, dim3(pl.nbm * pl.nbn, pl.ksplit), dim3(256 * KG), lds, st, p);
    return hipGetLastError();
}

template <int BITS, typename T, int MT, int UNR>
static hipError_t launch_skinny_one(const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    const size_t lds = (size_t)pl.waves * 8192;
    hipLaunchKernelGGL((gemm_skinny_kernel<BITS, T, MT, UNR>), dim3(pl.nbn, pl.ksplit, pl.nbm), dim3(pl.waves * 64), lds, st, p);
    return hipGetLastError();
}

template <typename T, int RT, int U>
static hipError_t launch_strip16_one(const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    const size_t lds = (size_t)pl.waves * RT * 1024;
    hipLaunchKernelGGL((gemm_strip16_kernel<T, RT, U>), dim3(pl.nbn, pl.ksplit, pl.nbm), dim3(pl.waves * 64), lds, st, p);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_strip16(const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    switch (pl.mt) {
        case 1: return launch_strip16_one<T, 1, 8>(pl, p, st);
        case 2: return launch_strip16_one<T, 2, 4>(pl, p, st);
        case 4: return launch_strip16_one<T, 4, 2>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

template <int BITS, typename T>
static hipError_t launch_skinny(const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    const bool u4 = pl.bk == 64;
    switch (pl.mt) {
        case 1: return u4 ? launch_skinny_one<BITS, T, 1, 4>(pl, p, st) : launch_skinny_one<BITS, T, 1, 2>(pl, p, st);
        case 2: return u4 ? launch_skinny_one<BITS, T, 2, 4>(pl, p, st) : launch_skinny_one<BITS, T, 2, 2>(pl, p, st);
        case 4:      // (3 bits, 64 rows: four unrolled steps spill 3..9 registers -- two at a time, which every group that allows four allows too)
            if constexpr (BITS == 3) return launch_skinny_one<BITS, T, 4, 2>(pl, p, st);
            else return u4 ? launch_skinny_one<BITS, T, 4, 4>(pl, p, st) : launch_skinny_one<BITS, T, 4, 2>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

template <int BITS, typename T>
static hipError_t launch_bits(const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    if constexpr (BITS == 4) {
        if (pl.strip16) return launch_strip16<T>(pl, p, st);
    }
    if (pl.skinny) return launch_skinny<BITS, T>(pl, p, st);
    if constexpr (BITS == 4) {
        if (pl.bk == 64) {
            if (pl.kg == 2) {
#ifdef GPTQ_GEMM_ABLATIONS
                if constexpr (std::is_same_v<T, f16>) {
                    if (pl.variant == 17) return launch_one<BITS, T, 4, 64, 16, false, false, 2>(pl, p, st);    // s_memtime timeline
                    if (pl.variant == 32) return (pl.xslot && pl.glds) ? launch_one<BITS, T, 4, 64, 32, true, true, 2>(pl, p, st)
                                                                       : launch_one<BITS, T, 4, 64, 32, false, false, 2>(pl, p, st);   // ping-pong
                }
#endif
                if constexpr (std::is_same_v<T, f16>) {
                    if (pl.xslot && pl.glds) return launch_one<BITS, T, 4, 64, 1, true, true, 2>(pl, p, st);
                }
                return launch_one<BITS, T, 4, 64, 1, false, false, 2>(pl, p, st);
            }
            if constexpr (std::is_same_v<T, f16>) {
                if (pl.variant == 3) return (pl.xslot && pl.glds) ? launch_one<BITS, T, 4, 64, 3, true, true>(pl, p, st)
                                                                  : launch_one<BITS, T, 4, 64, 3>(pl, p, st);   // experiment: cross-step pipeline
                if (pl.xslot && pl.glds) return launch_one<BITS, T, 4, 64, 1, true, true>(pl, p, st);
                if (pl.xslot) return launch_one<BITS, T, 4, 64, 1, true>(pl, p, st);
                if (pl.variant == 1) return launch_one<BITS, T, 4, 64, 0>(pl, p, st);   // experiment: plain loop
                if (pl.variant == 2) return launch_one<BITS, T, 4, 64, 2>(pl, p, st);   // experiment: plain + setprio
#ifdef GPTQ_GEMM_ABLATIONS
                if (pl.variant == 9) return launch_one<BITS, T, 4, 64, 9>(pl, p, st);
                if (pl.variant == 10) return launch_one<BITS, T, 4, 64, 10>(pl, p, st);
                if (pl.variant == 11) return launch_one<BITS, T, 4, 64, 11>(pl, p, st);
                if (pl.variant == 12) return launch_one<BITS, T, 4, 64, 12>(pl, p, st);
                if (pl.variant == 15) return launch_one<BITS, T, 4, 64, 15>(pl, p, st);
                if (pl.variant == 16) return launch_one<BITS, T, 4, 64, 16>(pl, p, st);                               // s_memtime timeline
                if (pl.variant == 24) return launch_one<BITS, T, 4, 64, 24>(pl, p, st);                               // loads interleaved with the MFMA groups
#endif
            }
            return launch_one<BITS, T, 4, 64>(pl, p, st);
        }
    }
    switch (pl.mt) {
        case 1: return launch_one<BITS, T, 1, 32>(pl, p, st);
        case 2: return launch_one<BITS, T, 2, 32>(pl, p, st);
        case 4: return launch_one<BITS, T, 4, 32>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

template <typename T>
static hipError_t launch_t(const gptq_layer_t& L, const GemmPlan& pl, const GemmParams& p, hipStream_t st) {
    switch (L.bits) {
        case 2: return launch_bits<2, T>(pl, p, st);
        case 3: return launch_bits<3, T>(pl, p, st);
        case 4: return launch_bits<4, T>(pl, p, st);
        case 8: return launch_bits<8, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

// x_permuted: the permuted x of this plan already sits at the head of `workspace` (an earlier layer of the same gptq_forward_multi call that shares
// this layer's perm put it there): the permute launch is skipped.
hipError_t launch_gemm(const gptq_layer_t& L, const GemmPlan& pl, const void* x, void* out, int M,
                       void* ws_header, void* workspace, hipStream_t st, bool x_permuted) {
    if (!pl.supported) return hipErrorNotSupported;
    GemmParams p{};
    p.qweight = pl.use_seq ? L.qweight_seq : L.qweight;
    p.qzeros = L.qzeros;
    p.scales = L.scales;
    p.bias = L.bias;
    p.x = x;
    p.out = out;
    p.M = M; p.K = L.K; p.N = L.N; p.group_size = L.group_size; p.zero_mode = L.zero_mode;
    p.nbm = pl.nbm; p.nbn = pl.nbn; p.ksplit = pl.ksplit;
    p.ksteps_total = pl.ksteps_total; p.ksteps_per_split = pl.ksteps_per_split;
    p.qrows = L.K / 32 * L.bits;
    if (!pl.f32) {
        const unsigned long long kpg = (unsigned long long)(L.group_size >= pl.bk ? L.group_size / pl.bk : 1);       // K-steps per group (the stream-K kernel alone runs 64-deep steps on 32-wide groups and has its own parameters)
        p.kpg_inv = ((1ull << 32) + kpg - 1) / kpg;
    }
    hipError_t e;
    if (pl.f32) {
        if (pl.use_seq) {
            e = launch_permute_columns(x, L.perm, M, L.K, GPTQ_F32, workspace, st);
            if (e != hipSuccess) return e;
            p.x = workspace;
        }
        const dim3 grid(pl.nbm * pl.nbn), block(256);
        switch (L.bits) {
            case 2: hipLaunchKernelGGL(gemm_f32_kernel<2>, grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL(gemm_f32_kernel<3>, grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL(gemm_f32_kernel<4>, grid, block, 0, st, p); break;
            case 8: hipLaunchKernelGGL(gemm_f32_kernel<8>, grid, block, 0, st, p); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (pl.use_seq) {
        // k-slot order for the DMA-staged tiled kernel: the LDS-staged row kernel; plain order: whichever permute kernel fits M and K
        // (a few long rows -- batched decode -- take the flat gather: 11008x4096 M = 8 act-order 16.8 -> ~14.5 us)
        if (!x_permuted) {
            e = pl.xnat ? launch_permute_rows16(x, L.perm, M, L.K, workspace, st, false)
                : pl.xslot ? launch_permute_rows16(x, L.perm, M, L.K, workspace, st, true)
                           : launch_permute_columns(x, L.perm, M, L.K, L.dtype, workspace, st);
            if (e != hipSuccess) return e;
        }
        p.x = workspace;
    }
    p.partial = (float*)((char*)workspace + pl.xperm_bytes);
    if (pl.tail > 0) {
        if (!ws_header || !workspace) return hipErrorInvalidValue;
        p.tail = pl.tail; p.tail_lg = pl.tail_lg;
        p.tail_flags = (unsigned*)ws_header;
        p.tail_partial = p.partial;
        p.err = (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2;
        p.max_spins = 1u << 22;
    }
    if (pl.mid) {
        const gptq_layer_t* one[1] = {&L};
        void* outs[1] = {out};
        return launch_mid(one, pl.midp, p.x, outs, M, ws_header, p.partial, pl.use_seq ? L.qweight_seq : nullptr, st);
    }
    if (pl.stream64) {
        const gptq_layer_t* one[1] = {&L};
        void* outs[1] = {out};
        Stream64Plan sp = plan_stream64(one, 1, M, nullptr);
        sp.waves = pl.waves; sp.u = pl.u; sp.ksplit = pl.ksplit; sp.ksteps_per_split = pl.ksteps_per_split;      // the (possibly tuned) geometry of plan_gemm
        const size_t land = (size_t)sp.waves * sp.u * 1024, slabs = (size_t)sp.waves * sp.mt * 4096;
        sp.lds_bytes = (land > slabs ? land : slabs) + 16;
        return launch_stream64(one, sp, p.x, outs, M, ws_header, p.partial, pl.use_seq ? L.qweight_seq : nullptr, st);
    }
    if (pl.rows) return launch_gemm_rows(L, pl.rowsp, p.x, out, M, st);
    if (pl.panel) return launch_gemm_panel(L, pl.panelp, p.x, out, M, st);
    if (pl.wsk) return launch_gemm_wide_sk(L, p.x, out, M, ws_header, (char*)workspace + pl.xperm_bytes, st);
    if (pl.wide) return launch_gemm_wide(L, p.qweight, p.x, out, M, pl.use_seq, st, pl.wide_tiled);
    e = (L.dtype == GPTQ_F16) ? launch_t<f16>(L, pl, p, st) : launch_t<bf16>(L, pl, p, st);
    if (e != hipSuccess) return e;
    if (pl.ksplit > 1) {
        const size_t total4 = (size_t)M * L.N / 4;
        int blocks = (int)((total4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (L.dtype == GPTQ_F16)
            hipLaunchKernelGGL(gemm_reduce_kernel<f16>, dim3(blocks), dim3(256), 0, st, p.partial, (const f16*)L.bias, (f16*)out, pl.ksplit, M, L.N);
        else
            hipLaunchKernelGGL(gemm_reduce_kernel<bf16>, dim3(blocks), dim3(256), 0, st, p.partial, (const bf16*)L.bias, (bf16*)out, pl.ksplit, M, L.N);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace gptq
