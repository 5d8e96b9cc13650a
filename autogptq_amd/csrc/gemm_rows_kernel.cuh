// gemm_rows_kernel.cuh -- the exchange-free batched-decode kernel (see gemm_rows.hip for the design) and the per-width launch ladder; instantiated by
// gemm_rows.hip (4 bits) and gemm_rows_b38.hip (3 / 8 bits).
#pragma once
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"
#include "launch.h"

namespace gptq {
namespace rowsk {

// 1 .. 4 layers that read the same x (q|k|v, gate|up: gptq_forward_multi) in ONE launch: the strip groups of layer 0, then of layer 1, ...
struct RowsSeg {
    const unsigned* qweight;      // the layer's decode copy (qweight_tiled)
    const char* qconst;           // its constant records (qconst_tiled): [strip][group][48 / 64 bytes]
    const void* bias;
    void* out;
    int N, strips;                // out_features, N / 16
};
struct RowsParams {
    RowsSeg seg[4];
    int sg_end[4];                // strip groups up to and including layer i (unused entries: INT_MAX)
    const void* x;                // [M][K] (act-order layers: permuted in natural order of the re-sequenced rows by the pre-pass)
    int M, K;
    int chunks;                   // K / 128
    int groups;
    int gshift;                   // group_size >= 128: group of chunk c = c >> gshift (31: one group)
    int npm;                      // row tiles of 16 RB rows
    int nsg;                      // strip groups of all layers
    int cpw;                      // chunks per wave (the last waves may run short or empty)
};

template <typename T> struct Mma16;
template <> struct Mma16<f16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// one column's constants, one word (8 weights in pair order) at a time: w - z exactly in packed fp16, times the scale with ONE rounding to T
template <typename T> struct Deq1;
template <> struct Deq1<f16> {
    f16x2 s2, c1, c2;
    __device__ __forceinline__ void setup(unsigned sraw, unsigned z) {
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        s2 = as_f16x2(sraw * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);                // -(1024 + z)
        c2 = c1 + k960;                                               // -(64 + z)
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2(wide::and_or(q, 0x000f000fu, 0x64006400u)) + c1;
        const f16x2 h1 = as_f16x2(wide::and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2;
        const f16x2 h2 = as_f16x2(wide::and_or(q8, 0x000f000fu, 0x64006400u)) + c1;
        const f16x2 h3 = as_f16x2(wide::and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2;
        return u32x4{wide::f16x2_bits(h0 * s2), wide::f16x2_bits(h1 * s2), wide::f16x2_bits(h2 * s2), wide::f16x2_bits(h3 * s2)};
    }
};
template <> struct Deq1<bf16> {
    float s;
    f16x2 c1, c2;
    __device__ __forceinline__ void setup(unsigned sraw, unsigned z) {
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sraw);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        c2 = c1 + k960;
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2(wide::and_or(q, 0x000f000fu, 0x64006400u)) + c1;
        const f16x2 h1 = as_f16x2(wide::and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2;
        const f16x2 h2 = as_f16x2(wide::and_or(q8, 0x000f000fu, 0x64006400u)) + c1;
        const f16x2 h3 = as_f16x2(wide::and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2;
        return u32x4{wide::bf16_scaled_pair(h0, s), wide::bf16_scaled_pair(h1, s), wide::bf16_scaled_pair(h2, s), wide::bf16_scaled_pair(h3, s)};
    }
};

// 3 bits (utils.hip prepack_decode_weights_kernel<3>): word j of the lane's three holds pairs 5 j + i at bit 3 i of its halves, bit 15 / 31 = bit j of k30 / k31;
// fields inside the fp16 mantissa are read in place (gemm_wide_common.cuh: Deq3 is the four-column form of the same arithmetic)
template <typename T> struct Deq1_3 {
    wide::Scale4<T> sc;                                            // (column 0 of it)
    f16x2 c0, c1, c2;
    __device__ __forceinline__ void setup(unsigned sraw, unsigned z) {
        sc.setup(u32x2{sraw & 0xffffu, 0u});
        c0 = as_f16x2(z * 0x00010001u + 0xE400E400u);                 // -(1024 + z)
        c1 = as_f16x2(z * 0x00080008u + 0xD800D800u);                 // -(128 + z)
        c2 = as_f16x2(z * 0x00400040u + 0xCC00CC00u);                 // -(16 + z)
    }
    __device__ __forceinline__ f16x2 p0(unsigned q) const { return as_f16x2(wide::and_or(q, 0x00070007u, 0x64006400u)) + c0; }
    __device__ __forceinline__ f16x2 p1(unsigned q) const {
        const f16x2 rr = {(f16)0.125f, (f16)0.125f};
        return as_f16x2(wide::and_or(q, 0x00380038u, 0x64006400u)) * rr + c1;
    }
    __device__ __forceinline__ f16x2 p2(unsigned q) const {
        const f16x2 rr = {(f16)0.015625f, (f16)0.015625f};
        return as_f16x2(wide::and_or(q, 0x01c001c0u, 0x64006400u)) * rr + c2;
    }
    __device__ __forceinline__ u32x4 frag(const wide::u32x3& w, int ks) const {      // k = 8 ks .. 8 ks + 7 of the lane's 32
        f16x2 h[4];
        if (ks == 0) {
            h[0] = p0(w[0]); h[1] = p1(w[0]); h[2] = p2(w[0]); h[3] = p1(w[0] >> 6);
        } else if (ks == 1) {
            h[0] = p2(w[0] >> 6); h[1] = p0(w[1]); h[2] = p1(w[1]); h[3] = p2(w[1]);
        } else if (ks == 2) {
            const unsigned q6 = w[1] >> 6;
            h[0] = p1(q6); h[1] = p2(q6); h[2] = p0(w[2]); h[3] = p1(w[2]);
        } else {
            const unsigned q6 = w[2] >> 6;
            unsigned t = (w[0] >> 15) & 0x00010001u;
            t = wide::and_or(w[1] >> 14, 0x00020002u, t);
            t = wide::and_or(w[2] >> 13, 0x00040004u, t);
            h[0] = p2(w[2]); h[1] = p1(q6); h[2] = p2(q6); h[3] = p0(t);
        }
        return u32x4{sc.mul(h[0], 0), sc.mul(h[1], 0), sc.mul(h[2], 0), sc.mul(h[3], 0)};
    }
};
// 8 bits: stored byte p of word w = k 4 w + {0, 2, 1, 3}[p]
template <typename T> struct Deq1_8 {
    wide::Scale4<T> sc;
    f16x2 c1;
    __device__ __forceinline__ void setup(unsigned sraw, unsigned z) {
        sc.setup(u32x2{sraw & 0xffffu, 0u});
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
    }
    __device__ __forceinline__ u32x4 frag(unsigned q0, unsigned q1) const {
        const f16x2 h0 = as_f16x2(wide::and_or(q0, 0x00ff00ffu, 0x64006400u)) + c1;
        const f16x2 h1 = as_f16x2(wide::and_or(q0 >> 8, 0x00ff00ffu, 0x64006400u)) + c1;
        const f16x2 h2 = as_f16x2(wide::and_or(q1, 0x00ff00ffu, 0x64006400u)) + c1;
        const f16x2 h3 = as_f16x2(wide::and_or(q1 >> 8, 0x00ff00ffu, 0x64006400u)) + c1;
        return u32x4{sc.mul(h0, 0), sc.mul(h1, 0), sc.mul(h2, 0), sc.mul(h3, 0)};
    }
};
template <typename T, int BITS> struct DeqSel { typedef Deq1<T> type; };
template <typename T> struct DeqSel<T, 3> { typedef Deq1_3<T> type; };
template <typename T> struct DeqSel<T, 8> { typedef Deq1_8<T> type; };

// GM: 0 = group_size a multiple of 128 (one group per chunk), 1 = 64 (k-slots 0, 1 | 2, 3), 2 = 32 (one group per k-slot)
// BITS = 8: the copy's chunks are 64 deep (a lane = 16 k of one column): a 128-deep x chunk takes two of them, MFMA step w reads words 2 (w & 1), + 1 of
// half w >> 1, and the constants are loaded per half (group modes: 128-multiples, 64 = one group per half, 32 = k-slots 0, 1 | 2, 3 of each half)
// Two x buffers per wave in LDS: the next chunk's rows are in flight under this chunk's MFMAs.  (One buffer and twice the waves: no faster; TWO chunks of x in
// flight per wave -- chunk c + 2 DMA'd into the buffer whose fragments have just gone to registers: parity-green and 3 - 8 % SLOWER in a same-session A/B of the
// two libraries: the pull is not bound by bytes in flight.  tools/rows_ab.py, profiles/r05_rows_ab.log.)
template <typename T, int BITS, int RB, int S, int GM>
__global__ void __launch_bounds__(RB == 2 ? 512 : 1024) gemm_rows_kernel(RowsParams p) {
    constexpr int XBUFS = 2;
    constexpr int NH = BITS == 8 ? 2 : 1;                      // chunks of the copy per 128-deep x chunk
    constexpr unsigned REC = BITS == 8 ? 64u : 48u, WCH = BITS == 3 ? 768u : 1024u;      // constant record, strip-chunk of the copy
    constexpr int R = 16 * RB, XB = R * 256, NDMA = R / 4, NW = 3 * NH * S;      // rows, bytes of one x chunk, its DMA instructions, a chunk's weight + constant loads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pm = Lb % p.npm, sgi = Lb / p.npm;             // consecutive workgroups (one XCD): the same strips, the next rows
    const int li = (sgi >= p.sg_end[0]) + (sgi >= p.sg_end[1]) + (sgi >= p.sg_end[2]);      // the layer this strip group belongs to
    const RowsSeg& Ls = p.seg[li];
    const int m0 = pm * R, s0 = (sgi - (li ? p.sg_end[li - 1] : 0)) * S;
    const int n_strips = Ls.strips, n_out = Ls.N;
    const char* const qw_base = (const char*)Ls.qweight;
    const char* const qc_base = Ls.qconst;
    const int r = lane & 15, g = lane >> 4;
    char* const xbuf = smem + (size_t)wave * XBUFS * XB;
    const unsigned xbuf_lds = lds_addr_of(xbuf);

    // x DMA i (4 rows x 256 bytes): lane (rr = lane >> 4, slot = lane & 15) fetches piece slot ^ (row & 15) of row 4 i + rr -- the four lanes of a quad stay inside
    // one 64-byte line (16 lines per instruction: the minimum) -- so that LDS slot s of row R holds piece s ^ (R & 15): the reads below (16 rows, one piece
    // each) then fall on 16 different slots = all 64 banks
    unsigned xoff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int row = 4 * i + g;
        const int m = min(m0 + row, p.M - 1);                 // rows past M repeat the last one (their outputs are not stored)
        xoff[i] = (unsigned)m * (unsigned)p.K * 2u + (unsigned)((r ^ (row & 15)) * 16);
    }
    // A fragment of row block rb, MFMA step w: the 16-byte piece of row 16 rb + r that holds the lane's k -- 3 / 4 bits: k = 32 g + 8 w (piece 4 g + w);
    // 8 bits: k = 64 (w >> 1) + 16 g + 8 (w & 1) (piece 8 (w >> 1) + 2 g + (w & 1)) -- stored at slot piece ^ r
    unsigned aoff[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int piece = BITS == 8 ? 8 * (w >> 1) + 2 * g + (w & 1) : 4 * g + w;
        aoff[w] = (unsigned)(r * 256 + ((piece ^ r) * 16));
    }
    // weights: lane * 16 (12 at 3 bits) of the strip-chunk; constants: record (strip, group): scale at 2 col, zero-point (as used) at 32 + col (8 bits: 32 + 2 col)
    const unsigned wlane = (unsigned)lane * (BITS == 3 ? 12u : 16u);
    const unsigned glane = BITS == 8 ? (GM == 2 ? (unsigned)(g >> 1) * REC : 0u) : (GM == 2 ? (unsigned)g * REC : (GM == 1 ? (unsigned)(g >> 1) * REC : 0u));
    const unsigned slane = glane + (unsigned)r * 2u, zlane = glane + 32u + (unsigned)r * (BITS == 8 ? 2u : 1u);
    const int c0 = wave * p.cpw, c1 = min(c0 + p.cpw, p.chunks);

    // One chunk's packed weights and constants (registers: every use below is inlined and unrolled).  The weights come from HBM, x from L2: a ring of DW
    // chunks of weights is kept in flight per wave (issued up front, refilled behind each chunk's MFMAs).
    constexpr int RING4 = 4 * S * (BITS == 8 ? 12 : (BITS == 3 ? 5 : 6));      // registers of a four-chunk ring
    constexpr int DW = RING4 <= (RB == 1 ? 56 : 120) ? 4 : 2;
    using WV = std::conditional_t<BITS == 3, wide::u32x3, u32x4>;      // a lane's words of one chunk of the copy (written by ONE load: no copies between the load and the wait)
    struct Buf { WV wq[S][NH]; unsigned cs[S][NH], cz[S][NH]; };
    Buf q[DW];
    auto issue_w = [&](int c, Buf& B) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int strip = min(s0 + s, n_strips - 1);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const char* wsrc = qw_base + ((size_t)strip * (p.chunks * NH) + (c * NH + h)) * WCH;
                const int grp = GM == 0 ? min(c >> p.gshift, p.groups - 1) : (BITS == 8 ? (GM == 1 ? 2 * c + h : 4 * c + 2 * h) : (GM == 1 ? 2 * c : 4 * c));
                const char* csrc = qc_base + ((size_t)strip * p.groups + grp) * REC;
                if constexpr (BITS == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(B.wq[s][h]) : "v"(wlane), "s"(wsrc) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(B.wq[s][h]) : "v"(wlane), "s"(wsrc) : "memory");
                asm volatile("global_load_ushort %0, %1, %2" : "=v"(B.cs[s][h]) : "v"(slane), "s"(csrc) : "memory");
                if constexpr (BITS == 8) asm volatile("global_load_ushort %0, %1, %2" : "=v"(B.cz[s][h]) : "v"(zlane), "s"(csrc) : "memory");
                else asm volatile("global_load_ubyte %0, %1, %2" : "=v"(B.cz[s][h]) : "v"(zlane), "s"(csrc) : "memory");
            }
        }
    };
    // DMAs [i0, i1) of chunk c's rows into buffer b
    auto issue_x = [&](int c, int b, int i0, int i1) __attribute__((always_inline)) {
        const char* xsrc = (const char*)p.x + (size_t)c * 256;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(xbuf_lds + (unsigned)(b * XB));
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const unsigned xo = xoff[i];                       // (an asm operand alone does not capture the array in a generic lambda)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (unsigned)(i * 1024)), "v"(xo), "s"(xsrc) : "memory");
        }
    };
    // the registers pass through a statement behind the wait so that no use of them is scheduled in front of it
    auto claim = [&](Buf& B) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int h = 0; h < NH; ++h) asm volatile("" : "+v"(B.wq[s][h]), "+v"(B.cs[s][h]), "+v"(B.cz[s][h])::"memory");
    };

    f32x4 acc[RB][S];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) acc[rb][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // chunk in buffer b; the NEXT chunk's x DMAs (cn, into the other buffer) are issued a quarter at a time behind the MFMAs of each step: eight DMAs in a row
    // fill the wave's VMEM queue and stall it at the issue port for as long as the TA takes to drain them -- the pull time then ADDS to the dequant time
    // instead of hiding under it (first version: time = 3.0 + 0.012 per KiB of x + 0.034 per KiB of weights, strictly additive; tools/rows_ab.py)
    auto compute = [&](const Buf& B, int b, int cn, bool has_x) __attribute__((always_inline)) {
        typename DeqSel<T, BITS>::type dq[S][NH];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int h = 0; h < NH; ++h) dq[s][h].setup(B.cs[s][h], B.cz[s][h]);
        const char* xb = xbuf + b * XB;
        u32x4 a[4][RB];                                        // all the chunk's x fragments first: the LDS latency lies under the dequant of step 0
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[w][rb] = *(const u32x4*)(xb + rb * 4096 + aoff[w]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                u32x4 bq;
                if constexpr (BITS == 4) bq = dq[s][0].frag(B.wq[s][0][w]);
                else if constexpr (BITS == 3) bq = dq[s][0].frag(B.wq[s][0], w);
                else bq = dq[s][w >> 1].frag(B.wq[s][w >> 1][2 * (w & 1)], B.wq[s][w >> 1][2 * (w & 1) + 1]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][s] = Mma16<T>::run(a[w][rb], bq, acc[rb][s]);
            }
            if (has_x) issue_x(cn, b ^ 1, w * (NDMA / 4), (w + 1) * (NDMA / 4));
        }
    };

    if (c0 < c1) {
#pragma unroll
        for (int j = 0; j < DW; ++j)
            if (c0 + j < c1) issue_w(c0 + j, q[j]);
        issue_x(c0, 0, 0, NDMA);
        for (int cb = c0; cb < c1; cb += DW) {
#pragma unroll
            for (int j = 0; j < DW; ++j) {
                const int c = cb + j;
                if (c >= c1) break;
                // VMEM queue, oldest first: W(c0 .. c0 + DW - 1), x(c0) | x(c0 + 1) under the MFMAs of c0, W(c0 + DW) | x(c0 + 2), W(c0 + DW + 1) | ...
                // chunk c needs x(c) and everything older (W(c) is); behind x(c) there is only W(c + DW - 1), issued at the end of the previous iteration
                const bool has_w = c > c0 && c + DW - 1 < c1;
                if (has_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                claim(q[j]);
                compute(q[j], j & 1, c + 1, c + 1 < c1);
                if (c + DW < c1) issue_w(c + DW, q[j]);
            }
        }
    }

    // ---- the waves' sums meet in LDS (fixed order), bias, store ------------------------------------------------------------------------------
    __syncthreads();                                           // every wave is done with its x buffers
    float* const red = (float*)smem;                           // [wave][rb][s][lane] float4
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) *(f32x4*)(red + ((size_t)((wave * RB + rb) * S + s) * 64 + lane) * 4) = acc[rb][s];
    __syncthreads();
    for (int item = tid; item < RB * S * 64; item += (int)blockDim.x) {
        const int l = item & 63, t = item >> 6, s = t % S, rb = t / S;
        f32x4 v = *(const f32x4*)(red + ((size_t)((0 * RB + rb) * S + s) * 64 + l) * 4);
        for (int w0 = 1; w0 < nw; w0 += 4) {                   // four loads in flight, added in wave order (a serial chain of 15 LDS round trips otherwise)
            f32x4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = *(const f32x4*)(red + ((size_t)((min(w0 + j, nw - 1) * RB + rb) * S + s) * 64 + l) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w0 + j < nw) v += t[j];
        }
        const int strip = s0 + s;
        if (strip >= n_strips) continue;
        const int n = strip * 16 + (l & 15);
        const float bv = Ls.bias ? DType<T>::to_f32(((const T*)Ls.bias)[n]) : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                          // C/D layout of the 16x16 MFMA: row = 4 (lane >> 4) + i, column = lane & 15
            const int m = m0 + 16 * rb + 4 * (l >> 4) + i;
            if (m < p.M) ((T*)Ls.out)[(size_t)m * n_out + n] = DType<T>::from_f32(v[i] + bv);
        }
    }
}

// ---- 64 rows per workgroup (4 bits) -------------------------------------------------------------------------------------------------------------
// The x an XCD's L2 hands out depends only on the number of strip groups (each pulls all M rows), the dequant work on the number of ROW TILES (each
// dequantises its strips again).  Up to 128 rows the 32-row form above is bound by the former (a 64-row form measured the same); from ~160 rows the row
// tiles multiply and this form -- 4 row blocks per workgroup, half the replication -- takes over from the older kernels up to where the stream-K prefill
// kernel starts.  64 rows x 128 k of x are 16 KiB -- two buffers per wave would leave LDS for 4 waves -- so a wave stages HALF chunks (64 k: 8 KiB, two
// buffers) and the k order changes with it: half h of a chunk is k-slots 2 h, 2 h + 1; MFMA step w of the half gives lane (r, g) the
// k = 32 (2 h + (g >> 1)) + 8 (2 (g & 1) + w) + 0..7 -- on the x side piece 2 g + w of the row's 128-byte half segment (contiguous: 8 rows per DMA
// instruction, a quad inside one line), on the weight side words 2 (g & 1) + {0, 1} of k-slot 2 h + (g >> 1) of the lane's column: ONE 8-byte load per half.
// The fragments of a half go to registers first; the SAME buffer then takes the next chunk's half under this half's MFMAs.
template <typename T, int S, int GM>
__global__ void __launch_bounds__(512) gemm_rows64_kernel(RowsParams p) {
    constexpr int RB = 4, R = 64, HB = R * 128, NDMA = 8, NW = 6 * S;      // bytes of one x half-chunk, its DMA instructions, a CHUNK's weight + constant loads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int pm = Lb % p.npm, sgi = Lb / p.npm;
    const int li = (sgi >= p.sg_end[0]) + (sgi >= p.sg_end[1]) + (sgi >= p.sg_end[2]);
    const RowsSeg& Ls = p.seg[li];
    const int m0 = pm * R, s0 = (sgi - (li ? p.sg_end[li - 1] : 0)) * S;
    const int n_strips = Ls.strips, n_out = Ls.N;
    const char* const qw_base = (const char*)Ls.qweight;
    const char* const qc_base = Ls.qconst;
    const int r = lane & 15, g = lane >> 4;
    char* const xbuf = smem + (size_t)wave * 2 * HB;
    const unsigned xbuf_lds = lds_addr_of(xbuf);

    // x DMA i of a half (8 rows x 128 bytes): lane (row 8 i + (lane >> 3), slot lane & 7) fetches piece slot ^ ((row >> 1) & 7): LDS slot s of row R holds piece
    // s ^ ((R >> 1) & 7), so the 16 rows of a fragment read (one piece each, 128-byte pitch) fall on all 64 banks
    unsigned xoff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int row = 8 * i + (lane >> 3);
        const int m = min(m0 + row, p.M - 1);
        xoff[i] = (unsigned)m * (unsigned)p.K * 2u + (unsigned)((((lane & 7) ^ ((row >> 1) & 7))) * 16);
    }
    unsigned aoff[2];                                          // fragment of row block rb, step w: piece 2 g + w of row 16 rb + r -> slot (2 g + w) ^ ((r >> 1) & 7)   (16 rb >> 1 is a multiple of 8)
#pragma unroll
    for (int w = 0; w < 2; ++w) aoff[w] = (unsigned)(r * 128 + (((2 * g + w) ^ ((r >> 1) & 7)) * 16));
    const unsigned wlane = (unsigned)(g >> 1) * 256u + (unsigned)r * 16u + (unsigned)(g & 1) * 8u;      // + 512 h: words 2 (g & 1), + 1 of k-slot 2 h + (g >> 1)
    const unsigned glane = GM == 2 ? (unsigned)(g >> 1) * 48u : 0u;      // 32-wide groups: group 4 c + 2 h + (g >> 1); 64-wide: 2 c + h; 128-multiples: c >> gshift
    const unsigned slane = glane + (unsigned)r * 2u, zlane = glane + 32u + (unsigned)r;
    const int c0 = wave * p.cpw, c1 = min(c0 + p.cpw, p.chunks);

    constexpr int DW = S <= 3 ? 4 : 2;
    struct Buf { u32x2 wq[S][2]; unsigned cs[S][2], cz[S][2]; };
    Buf q[DW];
    auto issue_w = [&](int c, Buf& B) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int strip = min(s0 + s, n_strips - 1);
            const char* wsrc = qw_base + ((size_t)strip * p.chunks + c) * 1024;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int grp = GM == 0 ? min(c >> p.gshift, p.groups - 1) : (GM == 1 ? 2 * c + h : 4 * c + 2 * h);
                const char* csrc = qc_base + ((size_t)strip * p.groups + grp) * 48;
                if (h == 0) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(B.wq[s][0]) : "v"(wlane), "s"(wsrc) : "memory");
                else asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "=v"(B.wq[s][1]) : "v"(wlane), "s"(wsrc) : "memory");
                asm volatile("global_load_ushort %0, %1, %2" : "=v"(B.cs[s][h]) : "v"(slane), "s"(csrc) : "memory");
                asm volatile("global_load_ubyte %0, %1, %2" : "=v"(B.cz[s][h]) : "v"(zlane), "s"(csrc) : "memory");
            }
        }
    };
    auto issue_x = [&](int c, int h, int i0, int i1) __attribute__((always_inline)) {      // DMAs [i0, i1) of half h of chunk c into buffer h
        const char* xsrc = (const char*)p.x + (size_t)c * 256 + (size_t)h * 128;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(xbuf_lds + (unsigned)(h * HB));
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const unsigned xo = xoff[i];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (unsigned)(i * 1024)), "v"(xo), "s"(xsrc) : "memory");
        }
    };
    auto claim = [&](Buf& B) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(B.wq[s][h]), "+v"(B.cs[s][h]), "+v"(B.cz[s][h])::"memory");
    };

    f32x4 acc[RB][S];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) acc[rb][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // half h of the chunk in buffer h; the same half of chunk cn follows it into the buffer under the MFMAs (its fragments are in registers by then)
    auto compute = [&](const Buf& B, int h, int cn, bool has_x) __attribute__((always_inline)) {
        Deq1<T> dq[S];
#pragma unroll
        for (int s = 0; s < S; ++s) dq[s].setup(B.cs[s][h], B.cz[s][h]);
        const char* xb = xbuf + h * HB;
        u32x4 a[2][RB];
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[w][rb] = *(const u32x4*)(xb + rb * 2048 + aoff[w]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the buffer is free from here on
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const u32x4 bq = dq[s].frag(B.wq[s][h][w]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][s] = Mma16<T>::run(a[w][rb], bq, acc[rb][s]);
            }
            if (has_x) issue_x(cn, h, w * (NDMA / 2), (w + 1) * (NDMA / 2));
        }
    };
    auto wait_vm = [&](bool dma, bool wts) __attribute__((always_inline)) {
        if (dma && wts) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NW) : "memory");
        else if (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else if (wts) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // VMEM queue of a wave, oldest first: W(c0 .. c0 + DW - 1), x(c0, 0), x(c0, 1) | x(c0 + 1, 0) under half 0 of c0, x(c0 + 1, 1) under half 1, W(c0 + DW) | ...
    //   half 0 of chunk c needs x(c, 0) (W(c) is older: DW >= 2); behind it: x(c, 1), then W(c - 1 + DW) (end of chunk c - 1, not for c = c0)
    //   half 1 needs x(c, 1); behind it: W(c - 1 + DW), x(c + 1, 0) (issued under half 0 of this chunk)
    if (c0 < c1) {
#pragma unroll
        for (int j = 0; j < DW; ++j)
            if (c0 + j < c1) issue_w(c0 + j, q[j]);
        issue_x(c0, 0, 0, NDMA);
        issue_x(c0, 1, 0, NDMA);
        for (int cb = c0; cb < c1; cb += DW) {
#pragma unroll
            for (int j = 0; j < DW; ++j) {
                const int c = cb + j;
                if (c >= c1) break;
                const bool has_w = c > c0 && c - 1 + DW < c1, more = c + 1 < c1;
                wait_vm(true, has_w);                          // x(c, 1) is always behind x(c, 0)
                claim(q[j]);
                compute(q[j], 0, c + 1, more);
                wait_vm(more, has_w);
                compute(q[j], 1, c + 1, more);
                if (c + DW < c1) issue_w(c + DW, q[j]);
            }
        }
    }

    __syncthreads();
    float* const red = (float*)smem;                           // [wave][rb][s][lane] float4
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < S; ++s) *(f32x4*)(red + ((size_t)((wave * RB + rb) * S + s) * 64 + lane) * 4) = acc[rb][s];
    __syncthreads();
    for (int item = tid; item < RB * S * 64; item += (int)blockDim.x) {
        const int l = item & 63, t = item >> 6, s = t % S, rb = t / S;
        f32x4 v = *(const f32x4*)(red + ((size_t)((0 * RB + rb) * S + s) * 64 + l) * 4);
        for (int w0 = 1; w0 < nw; w0 += 4) {
            f32x4 tt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) tt[j] = *(const f32x4*)(red + ((size_t)((min(w0 + j, nw - 1) * RB + rb) * S + s) * 64 + l) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w0 + j < nw) v += tt[j];
        }
        const int strip = s0 + s;
        if (strip >= n_strips) continue;
        const int n = strip * 16 + (l & 15);
        const float bv = Ls.bias ? DType<T>::to_f32(((const T*)Ls.bias)[n]) : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 16 * rb + 4 * (l >> 4) + i;
            if (m < p.M) ((T*)Ls.out)[(size_t)m * n_out + n] = DType<T>::from_f32(v[i] + bv);
        }
    }
}

}  // namespace rowsk

// ---- instantiation ladder of one bit width (grant the dynamic LDS, launch) ----------------------------------------------------------------------
template <typename T, int BITS, int RB, int S, int GM>
static hipError_t rows_grant_one() {
    return hipFuncSetAttribute((const void*)rowsk::gemm_rows_kernel<T, BITS, RB, S, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T, int BITS, int RB, int GM>
static hipError_t rows_grant_s() {
    hipError_t e = rows_grant_one<T, BITS, RB, 1, GM>();
    if (e == hipSuccess) e = rows_grant_one<T, BITS, RB, 2, GM>();
    if (e == hipSuccess) e = rows_grant_one<T, BITS, RB, 3, GM>();
    if constexpr (!(BITS == 8 && RB == 1)) { if (e == hipSuccess) e = rows_grant_one<T, BITS, RB, 4, GM>(); }
    if constexpr (RB == 2 && BITS == 4) { if (e == hipSuccess) e = rows_grant_one<T, BITS, RB, 6, GM>(); }      // (RB = 1: 6 strips spill at 128 registers; 8 spill in either form)
    return e;
}
template <typename T, int BITS>
static hipError_t rows_grant_t() {
    hipError_t e = rows_grant_s<T, BITS, 1, 0>();
    if (e == hipSuccess) e = rows_grant_s<T, BITS, 2, 0>();
    if (e == hipSuccess) e = rows_grant_s<T, BITS, 1, 1>();
    if (e == hipSuccess) e = rows_grant_s<T, BITS, 2, 1>();
    if (e == hipSuccess) e = rows_grant_s<T, BITS, 1, 2>();
    if (e == hipSuccess) e = rows_grant_s<T, BITS, 2, 2>();
    return e;
}
template <typename T, int GM> static hipError_t rows64_grant();
template <int BITS>
static hipError_t rows_grant_bits() {
    hipError_t e = rows_grant_t<f16, BITS>();
    if (e == hipSuccess) e = rows_grant_t<bf16, BITS>();
    if constexpr (BITS == 4) {
        if (e == hipSuccess) e = rows64_grant<f16, 0>();
        if (e == hipSuccess) e = rows64_grant<f16, 1>();
        if (e == hipSuccess) e = rows64_grant<f16, 2>();
        if (e == hipSuccess) e = rows64_grant<bf16, 0>();
        if (e == hipSuccess) e = rows64_grant<bf16, 1>();
        if (e == hipSuccess) e = rows64_grant<bf16, 2>();
    }
    return e;
}

template <typename T, int BITS, int RB, int S, int GM>
static void rows_launch_one(const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) {
    hipLaunchKernelGGL((rowsk::gemm_rows_kernel<T, BITS, RB, S, GM>), dim3(pl.npm * pl.nsg), dim3(pl.waves * 64), pl.lds_bytes, st, p);
}
template <typename T, int BITS, int RB, int GM>
static hipError_t rows_launch_s(const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) {
    switch (pl.s) {
        case 1: rows_launch_one<T, BITS, RB, 1, GM>(pl, p, st); break;
        case 2: rows_launch_one<T, BITS, RB, 2, GM>(pl, p, st); break;
        case 3: rows_launch_one<T, BITS, RB, 3, GM>(pl, p, st); break;
        case 4: if constexpr (!(BITS == 8 && RB == 1)) { rows_launch_one<T, BITS, RB, 4, GM>(pl, p, st); break; } else return hipErrorInvalidValue;      // (8 bits, one row block, four strips: spills -- plan_rows never asks for it)
        case 6: if constexpr (RB == 2 && BITS == 4) { rows_launch_one<T, BITS, RB, 6, GM>(pl, p, st); break; } else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <typename T, int GM>
static hipError_t rows64_launch(const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) {
    const dim3 grid(pl.npm * pl.nsg), block(pl.waves * 64);
    switch (pl.s) {
        case 1: hipLaunchKernelGGL((rowsk::gemm_rows64_kernel<T, 1, GM>), grid, block, pl.lds_bytes, st, p); break;
        case 2: hipLaunchKernelGGL((rowsk::gemm_rows64_kernel<T, 2, GM>), grid, block, pl.lds_bytes, st, p); break;
        case 3: hipLaunchKernelGGL((rowsk::gemm_rows64_kernel<T, 3, GM>), grid, block, pl.lds_bytes, st, p); break;
        case 4: hipLaunchKernelGGL((rowsk::gemm_rows64_kernel<T, 4, GM>), grid, block, pl.lds_bytes, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <typename T, int GM>
static hipError_t rows64_grant() {
    hipError_t e = hipFuncSetAttribute((const void*)rowsk::gemm_rows64_kernel<T, 1, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rowsk::gemm_rows64_kernel<T, 2, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rowsk::gemm_rows64_kernel<T, 3, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rowsk::gemm_rows64_kernel<T, 4, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}
template <typename T, int BITS, int GM>
static hipError_t rows_launch_rb(const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) {
    if constexpr (BITS == 4) {
        if (pl.rb == 4) return rows64_launch<T, GM>(pl, p, st);
    }
    return pl.rb == 1 ? rows_launch_s<T, BITS, 1, GM>(pl, p, st) : rows_launch_s<T, BITS, 2, GM>(pl, p, st);
}
template <int BITS>
static hipError_t rows_launch_bits(int dtype, int gm, const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) {
    if (dtype == GPTQ_F16) return gm == 0 ? rows_launch_rb<f16, BITS, 0>(pl, p, st) : (gm == 1 ? rows_launch_rb<f16, BITS, 1>(pl, p, st) : rows_launch_rb<f16, BITS, 2>(pl, p, st));
    return gm == 0 ? rows_launch_rb<bf16, BITS, 0>(pl, p, st) : (gm == 1 ? rows_launch_rb<bf16, BITS, 1>(pl, p, st) : rows_launch_rb<bf16, BITS, 2>(pl, p, st));
}

}  // namespace gptq
