// declab.hip -- round 4 decode lab: which per-lane decomposition of a strip-major weight copy gets closest to the bare stream?
// Not part of the product (timing only: the weights are random bytes, results are not checked here; the product kernels have the parity tests).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/declab.hip -o tools/declab
//
//   V1  = the first tiled kernel (csrc/gemv_tiled.hip as of its first GPU run): lane = 4 adjacent columns x 1 packed row; every row of a lane is its own
//         group -> one (scales, zeros) load pair + one constant set (20 + 12 VALU) per 32 weights, x as 16 bytes per row from L2, v_perm to slot order.
//   V2  = column-per-lane: the 1 KiB chunk (16 packed rows x 16 columns = one group of 128) is stored [k-slot 4][column 16][4 words]; lane = (k-slot, column)
//         owns 4 consecutive packed rows (32 k) of ONE column: one scale (2 B), one zero nibble, one constant set (5 + 2 VALU) per 32 weights; nibbles are
//         stored so that the magic-number extraction yields natural k pairs (no v_perm on x); x comes from LDS (staged once per workgroup) or from L2.
// Both: 16-column strips, one strip per workgroup over all K, W waves x U chunks in flight, k-reduction on v_mfma_f32_4x4x4, fp32 group sums x scale.
// ABL bits: 1 = no x loads, 2 = no math (xor of the loaded words), 4 = no scale / zero loads, 8 = no epilogue (no cross-lane / cross-wave reduction, no store).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x2 as_f16x2(unsigned v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ void dma16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ f32x4 mma4(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
}

struct P {
    const unsigned* tq;      // [strips][R][16] words (any inner order: timing only)
    const unsigned* qzeros;  // [G][N/8]
    const f16* scales;       // [G][N]
    const f16* x;            // [M][K]
    f16* out;                // [M][N]
    int M, K, N, R, gshift;  // R = K/8 packed rows, gshift = log2(rows per group)
};

// ------------------------------------------------------------------------------------------------ V1
template <int MT, int U, bool DMA, int ABL, int MAXW>
__global__ void __launch_bounds__(MAXW * 64, MAXW == 16 ? 4 : 2) dec_v1(P p) {
    unsigned m_lo, m_hi, magic;
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane & 3, rs = lane >> 2;
    char* const wq = smem + (size_t)wave * (U * 1024);
    const unsigned wq_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)wq);
    float* const red = (float*)(smem + (DMA ? (size_t)W * (U * 1024) : (size_t)0));
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int N = p.N, R = p.R, n0 = strip * 16 + cl * 4;
    const char* const xb = (const char*)p.x;
    const unsigned x_lane = (unsigned)min(lane & 3, p.M - 1) * (unsigned)p.K * 2u;
    const char* const sb = (const char*)p.scales;
    const unsigned s_lane = (unsigned)n0 * 2u, s_row = (unsigned)N * 2u;
    const char* const zb = (const char*)p.qzeros;
    const unsigned z_lane = (unsigned)(n0 >> 3) * 4u, z_row = (unsigned)(N >> 3) * 4u;
    const char* const tb = (const char*)(p.tq + (size_t)strip * R * 16);
    const unsigned t_lane = (unsigned)cl * 16u;
    const unsigned zsh = (unsigned)(n0 & 7) * 4u;
    float acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
    const f16x2 k960 = {(f16)960.f, (f16)960.f}, r16 = {(f16)0.0625f, (f16)0.0625f};
    for (int base = 0; base < R; base += W * U * 16) {
        const int r0 = base + wave * (U * 16) + rs;
        u32x2 sraw[U];
        unsigned zw[U];
        u32x4 xr[U], q[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const unsigned rc = (unsigned)min(r0 + j * 16, R - 1), g = rc >> p.gshift;
            if constexpr (ABL & 4) { sraw[j] = u32x2{0x14001400u + g, 0x14001400u}; zw[j] = 0x77777777u; }
            else { sraw[j] = *(const u32x2*)(sb + (g * s_row + s_lane)); zw[j] = *(const unsigned*)(zb + (g * z_row + z_lane)); }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if constexpr (ABL & 1) xr[j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (unsigned)r0};
            else xr[j] = *(const u32x4*)(xb + ((unsigned)min(r0 + j * 16, R - 1) * 16u + x_lane));
        }
        if constexpr (DMA) {
            if (base) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < U; ++j) dma16_nt(tb + ((unsigned)min(r0 + j * 16, R - 1) * 64u + t_lane), wq_lds + j * 1024);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) q[j] = __builtin_nontemporal_load((const u32x4*)(tb + ((unsigned)min(r0 + j * 16, R - 1) * 64u + t_lane)));
        }
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] {
                 constexpr int j = J;
                 u32x4 qv;
                 if constexpr (DMA) {
                     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1 - j) : "memory");
                     qv = *(const u32x4*)(wq + j * 1024 + lane * 16);
                 } else qv = q[j];
                 if constexpr (ABL & 2) {
#pragma unroll
                     for (int c = 0; c < 4; ++c) acc[c][0] += __builtin_bit_cast(float, (qv[c] ^ xr[j][c] ^ sraw[j][c >> 1] ^ zw[j]) & 0x3fffffffu);
                     return;
                 }
                 const bool live = (r0 + j * 16 < R);
                 const u32x4 t = xr[j];
                 u32x2 a01 = u32x2{__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u)};
                 u32x2 a23 = u32x2{__builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
                 if (!live) { a01 = u32x2{0u, 0u}; a23 = u32x2{0u, 0u}; }
                 const unsigned zw_ = zw[j] >> zsh;
                 f32x4 accg[4];
#pragma unroll
                 for (int c = 0; c < 4; ++c) {
                     const unsigned z = (((zw_ >> (4 * c)) & 15u) + 1u) & 15u;
                     const f16x2 c1 = as_f16x2(z * 0x00010001u + 0xE400E400u), c2 = c1 + k960;
                     const unsigned qw = qv[c], q8 = qw >> 8;
                     const f16x2 h0 = as_f16x2((qw & m_lo) | magic) + c1, h1 = as_f16x2((qw & m_hi) | magic) * r16 + c2;
                     const f16x2 h2 = as_f16x2((q8 & m_lo) | magic) + c1, h3 = as_f16x2((q8 & m_hi) | magic) * r16 + c2;
                     accg[c] = mma4(a01, u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)}, f32x4{0.f, 0.f, 0.f, 0.f});
                     accg[c] = mma4(a23, u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)}, accg[c]);
                 }
#pragma unroll
                 for (int c = 0; c < 4; ++c) {
                     const unsigned sw = sraw[j][c >> 1];
                     const float sc = (float)__builtin_bit_cast(f16, (unsigned short)((c & 1) ? (sw >> 16) : (sw & 0xffffu)));
#pragma unroll
                     for (int m = 0; m < MT; ++m) acc[c][m] = fmaf(sc, accg[c][m], acc[c][m]);
                 }
             }()),
             ...);
        }(std::make_integer_sequence<int, U>{});
    }
    if constexpr (ABL & 8) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m) t += acc[c][m];
        if (t == 1.2345f) p.out[n0] = (f16)t;
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[c][m];
            v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            acc[c][m] = v;
        }
    if (lane < 4) {
#pragma unroll
        for (int m = 0; m < MT; ++m) *(f32x4*)(red + (wave * MT + m) * 16 + lane * 4) = f32x4{acc[0][m], acc[1][m], acc[2][m], acc[3][m]};
    }
    __syncthreads();
    for (int e = tid; e < MT * 16; e += blockDim.x) {
        const int m = e >> 4, c = e & 15;
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += red[(w * MT + m) * 16 + c];
        if (m < p.M) p.out[(size_t)m * N + strip * 16 + c] = (f16)s;
    }
}

// ------------------------------------------------------------------------------------------------ V2
// XSRC: 0 = x from L2 (4 x 16 bytes per chunk and lane), 1 = x staged in LDS once per workgroup (ds_read_b128)
template <int MT, int U, bool DMA, int XSRC, int ABL, int MAXW>
__global__ void __launch_bounds__(MAXW * 64, MAXW == 16 ? 4 : 2) dec_v2(P p) {
    unsigned m_lo, m_hi, magic;
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int col = lane & 15, kb = lane >> 4;                       // lane = kb * 16 + col: lane-linear inside the chunk
    const int N = p.N, R = p.R, K = p.K;
    const int xstride = K * 2 + 16;                                   // LDS row stride of x (16-byte pad: the 4 rows of a 4-lane group hit different banks)
    char* const xs = smem;
    const size_t xbytes = XSRC == 1 ? (size_t)MT * xstride : 0;
    char* const wq = smem + xbytes + (size_t)wave * (U * 1024);
    const unsigned wq_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)wq);
    float* const red = (float*)(smem + xbytes + (DMA ? (size_t)W * (U * 1024) : (size_t)0));
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int n = strip * 16 + col;
    const int nchunks = R >> 4;                                       // lab: K % 128 == 0
    const char* const tb = (const char*)(p.tq + (size_t)strip * R * 16);
    const unsigned t_lane = (unsigned)lane * 16u;
    const char* const sb = (const char*)p.scales;
    const unsigned s_lane = (unsigned)n * 2u, s_row = (unsigned)N * 2u;
    const char* const zb = (const char*)p.qzeros;
    const unsigned z_lane = (unsigned)(n >> 3) * 4u, z_row = (unsigned)(N >> 3) * 4u, zsh = (unsigned)(n & 7) * 4u;
    const int xr_i = min(lane & 3, p.M - 1);                          // A operand: lane i of a 4-lane group carries x row i
    const char* const xg = (const char*)p.x + (size_t)xr_i * K * 2 + kb * 64;
    // stage x: 16-byte pieces, coalesced; the loads are issued HERE (first: loads return in issue order), the LDS writes behind the first weight burst
    constexpr int NP = 6;
    u32x4 xst[MT][NP];
    const int pieces = K / 8;
    if constexpr (XSRC == 1 && !(ABL & 1)) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int pc = tid + i * (int)blockDim.x;
                xst[m][i] = *(const u32x4*)((const char*)p.x + ((size_t)min(m, p.M - 1) * K + (size_t)min(pc, pieces - 1) * 8) * 2);
            }
    }
    const char* const xl = xs + (size_t)min(lane & 3, MT - 1) * xstride + kb * 64;
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    const f16x2 k960 = {(f16)960.f, (f16)960.f}, r16 = {(f16)0.0625f, (f16)0.0625f};
    bool staged = (XSRC != 1) || (ABL & 1);
    for (int cbase = 0; cbase < nchunks; cbase += W * U) {
        const int c0 = cbase + wave * U;
        unsigned short sraw[U];
        unsigned zw[U];
        u32x4 q[U];
        u32x4 xv[U][4];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const unsigned cc = (unsigned)min(c0 + j, nchunks - 1);
            const unsigned g = (cc * 16u + (unsigned)kb * 4u) >> p.gshift;
            if constexpr (ABL & 4) { sraw[j] = (unsigned short)(0x1400u + g); zw[j] = 0x77777777u; }
            else { sraw[j] = *(const unsigned short*)(sb + (g * s_row + s_lane)); zw[j] = *(const unsigned*)(zb + (g * z_row + z_lane)); }
        }
        if constexpr (XSRC == 0 && !(ABL & 1)) {
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int w = 0; w < 4; ++w) xv[j][w] = *(const u32x4*)(xg + ((unsigned)min(c0 + j, nchunks - 1) * 256u + w * 16u));
        }
        if constexpr (DMA) {
            if (cbase) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < U; ++j) dma16_nt(tb + ((unsigned)min(c0 + j, nchunks - 1) * 1024u + t_lane), wq_lds + j * 1024);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) q[j] = __builtin_nontemporal_load((const u32x4*)(tb + ((unsigned)min(c0 + j, nchunks - 1) * 1024u + t_lane)));
        }
        if (!staged) {                                                // first pass only (uniform): x to LDS behind the weight burst, then the one barrier
            if constexpr (XSRC == 1 && !(ABL & 1)) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const int pc = tid + i * (int)blockDim.x;
                        if (pc < pieces) *(u32x4*)(xs + (size_t)m * xstride + pc * 16) = xst[m][i];
                    }
                    for (int pc = tid + NP * (int)blockDim.x; pc < pieces; pc += (int)blockDim.x)     // long K with a small workgroup
                        *(u32x4*)(xs + (size_t)m * xstride + pc * 16) = *(const u32x4*)((const char*)p.x + ((size_t)min(m, p.M - 1) * K + (size_t)pc * 8) * 2);
                }
            }
            __syncthreads();
            staged = true;
        }
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] {
                 constexpr int j = J;
                 u32x4 qv;
                 if constexpr (DMA) {
                     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1 - j) : "memory");
                     qv = *(const u32x4*)(wq + j * 1024 + lane * 16);
                 } else qv = q[j];
                 u32x4 xa[4];
                 if constexpr (ABL & 1) {
#pragma unroll
                     for (int w = 0; w < 4; ++w) xa[w] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (unsigned)c0};
                 } else if constexpr (XSRC == 1) {
#pragma unroll
                     for (int w = 0; w < 4; ++w) xa[w] = *(const u32x4*)(xl + ((unsigned)min(c0 + j, nchunks - 1) * 256u + w * 16u));
                 } else {
#pragma unroll
                     for (int w = 0; w < 4; ++w) xa[w] = xv[j][w];
                 }
                 if constexpr (ABL & 2) {
#pragma unroll
                     for (int w = 0; w < 4; ++w) acc[0] += __builtin_bit_cast(float, (qv[w] ^ xa[w][0] ^ xa[w][3] ^ (unsigned)sraw[j] ^ zw[j]) & 0x3fffffffu);
                     return;
                 }
                 const bool live = (c0 + j < nchunks);
                 const unsigned z = (((zw[j] >> zsh) & 15u) + 1u) & 15u;
                 const f16x2 c1 = as_f16x2(z * 0x00010001u + 0xE400E400u), c2 = c1 + k960;
                 f32x4 accg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                 for (int w = 0; w < 4; ++w) {
                     const unsigned qw = qv[w], q8 = qw >> 8;
                     const f16x2 h0 = as_f16x2((qw & m_lo) | magic) + c1, h1 = as_f16x2((qw & m_hi) | magic) * r16 + c2;
                     const f16x2 h2 = as_f16x2((q8 & m_lo) | magic) + c1, h3 = as_f16x2((q8 & m_hi) | magic) * r16 + c2;
                     accg = mma4(u32x2{xa[w][0], xa[w][1]}, u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)}, accg);
                     accg = mma4(u32x2{xa[w][2], xa[w][3]}, u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)}, accg);
                 }
                 const float sc = live ? (float)__builtin_bit_cast(f16, sraw[j]) : 0.f;
#pragma unroll
                 for (int m = 0; m < MT; ++m) acc[m] = fmaf(sc, accg[m], acc[m]);
             }()),
             ...);
        }(std::make_integer_sequence<int, U>{});
    }
    if constexpr (ABL & 8) {
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) t += acc[m];
        if (t == 1.2345f) p.out[n] = (f16)t;
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float v = acc[m];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        acc[m] = v;
    }
    if (lane < 16) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[(wave * MT + m) * 16 + lane] = acc[m];
    }
    __syncthreads();
    for (int e = tid; e < MT * 16; e += blockDim.x) {
        const int m = e >> 4, c = e & 15;
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += red[(w * MT + m) * 16 + c];
        if (m < p.M) p.out[(size_t)m * N + strip * 16 + c] = (f16)s;
    }
}

// ------------------------------------------------------------------------------------------------ V3
// V2 (column per lane, x in LDS) + the group constants from a strip-major constants copy staged in LDS once per workgroup
// (cst[strip][g][48 B] = 16 fp16 scales + 16 one-byte zero-points as used) + S strips per workgroup that share every x fragment.
template <int MT, int U, int S, int ABL, int MAXW>
__global__ void __launch_bounds__(MAXW * 64, MAXW == 16 ? 4 : 2) dec_v3(P p, const unsigned char* cst) {
    static_assert(U % S == 0, "U loads in flight = U / S chunk positions x S strips");
    constexpr int UC = U / S;
    unsigned m_lo, m_hi, magic;
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int col = lane & 15, kb = lane >> 4;
    const int N = p.N, R = p.R, K = p.K;
    const int G = K >> 7;                                              // lab: group_size 128
    const int xstride = K * 2 + 16;
    char* const xs = smem;
    char* const cs = smem + (size_t)MT * xstride;                      // [S][G][48]
    float* const red = (float*)(cs + (size_t)S * G * 48);
    const int strip0 = xcd_remap(blockIdx.x, gridDim.x) * S;
    const int nchunks = R >> 4;
    const unsigned t_lane = (unsigned)lane * 16u;
    // stage x and the constants by LDS DMA (no VGPRs, issued FIRST: loads return in issue order), waited for behind the first weight burst
    const int pieces = K / 8;                                          // 16-byte pieces per x row
    const int cpieces = S * G * 3;                                     // ... of the S strips' constants (contiguous: the strips are adjacent)
    const char* const cg = (const char*)cst + (size_t)strip0 * G * 48;
    const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)xs, cs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cs;
    if constexpr (!(ABL & 1)) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            for (int pc0 = wave * 64; pc0 < pieces; pc0 += W * 64)          // wave-uniform trip count
                if (pc0 + lane < pieces)
                    dma16_nt((const char*)p.x + ((size_t)min(m, p.M - 1) * K + (size_t)(pc0 + lane) * 8) * 2, __builtin_amdgcn_readfirstlane(xs_lds + m * xstride + pc0 * 16));
    }
    if constexpr (!(ABL & 4)) {
        for (int pc0 = wave * 64; pc0 < cpieces; pc0 += W * 64)
            if (pc0 + lane < cpieces) dma16_nt(cg + (size_t)(pc0 + lane) * 16, __builtin_amdgcn_readfirstlane(cs_lds + pc0 * 16));
    }
    const char* const xl = xs + (size_t)min(lane & 3, MT - 1) * xstride + kb * 64;
    float acc[S][MT];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[s][m] = 0.f;
    const f16x2 k960 = {(f16)960.f, (f16)960.f}, r16 = {(f16)0.0625f, (f16)0.0625f};
    bool staged = false;
    for (int cbase = 0; cbase < nchunks; cbase += W * UC) {
        const int c0 = cbase + wave * UC;
        u32x4 q[UC][S];
#pragma unroll
        for (int j = 0; j < UC; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s)
                q[j][s] = __builtin_nontemporal_load((const u32x4*)((const char*)(p.tq + (size_t)(strip0 + s) * R * 16) + ((unsigned)min(c0 + j, nchunks - 1) * 1024u + t_lane)));
        if (!staged) {                                                // the staging DMAs are older than the U weight loads just issued
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
            __syncthreads();
            staged = true;
        }
#pragma unroll
        for (int j = 0; j < UC; ++j) {
            const int cc = min(c0 + j, nchunks - 1);
            const bool live = (c0 + j < nchunks);
            u32x4 xa[4];
            if constexpr (ABL & 1) {
#pragma unroll
                for (int w = 0; w < 4; ++w) xa[w] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (unsigned)c0};
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) xa[w] = *(const u32x4*)(xl + ((unsigned)cc * 256u + w * 16u));
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const char* cp = cs + ((size_t)s * G + cc) * 48;      // group = chunk (g128)
                unsigned short sraw; unsigned z;
                if constexpr (ABL & 4) { sraw = (unsigned short)(0x1400u + cc); z = 7u; }
                else { sraw = *(const unsigned short*)(cp + col * 2); z = *(const unsigned char*)(cp + 32 + col); }
                const u32x4 qv = q[j][s];
                if constexpr (ABL & 2) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) acc[s][0] += __builtin_bit_cast(float, (qv[w] ^ xa[w][0] ^ xa[w][3] ^ (unsigned)sraw ^ z) & 0x3fffffffu);
                    continue;
                }
                const f16x2 c1 = as_f16x2(z * 0x00010001u + 0xE400E400u), c2 = c1 + k960;
                f32x4 accg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned qw = qv[w], q8 = qw >> 8;
                    const f16x2 h0 = as_f16x2((qw & m_lo) | magic) + c1, h1 = as_f16x2((qw & m_hi) | magic) * r16 + c2;
                    const f16x2 h2 = as_f16x2((q8 & m_lo) | magic) + c1, h3 = as_f16x2((q8 & m_hi) | magic) * r16 + c2;
                    accg = mma4(u32x2{xa[w][0], xa[w][1]}, u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)}, accg);
                    accg = mma4(u32x2{xa[w][2], xa[w][3]}, u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)}, accg);
                }
                const float sc = live ? (float)__builtin_bit_cast(f16, sraw) : 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[s][m] = fmaf(sc, accg[m], acc[s][m]);
            }
        }
    }
    if constexpr (ABL & 8) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) t += acc[s][m];
        if (t == 1.2345f) p.out[strip0 * 16 + col] = (f16)t;
        return;
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v = acc[s][m];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            acc[s][m] = v;
        }
    if (lane < 16) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) red[((wave * S + s) * MT + m) * 16 + lane] = acc[s][m];
    }
    __syncthreads();
    for (int e = tid; e < S * MT * 16; e += blockDim.x) {
        const int c = e & 15, m = (e >> 4) % MT, s = (e >> 4) / MT;
        float t = 0.f;
        for (int w = 0; w < W; ++w) t += red[((w * S + s) * MT + m) * 16 + c];
        if (m < p.M) p.out[(size_t)m * N + (strip0 + s) * 16 + c] = (f16)t;
    }
}

// ------------------------------------------------------------------------------------------------ harness
struct Res { float us; char name[160]; };
static hipStream_t st;
static unsigned* wpool; static size_t wpool_bytes;
static f16 *scales_pool, *xbuf, *outbuf; static unsigned* zeros_pool;

template <typename F>
static float time_graph(F launch_one, size_t n, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (size_t i = 0; i < n; ++i) launch_one(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1e3f / n);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

struct Shape { int K, N; const char* what; };

template <typename KERN>
static void run_kernel(std::vector<Res>& res, KERN kern, const Shape& s, int M, int W, size_t lds, const char* name) {
    const int R = s.K / 8, G = s.K / 128;
    const size_t wbytes = (size_t)R * s.N * 4, sbytes = (size_t)G * s.N * 2, zbytes = (size_t)G * (s.N / 8) * 4;
    const size_t per = wbytes + ((sbytes + zbytes + 255) / 256) * 256;
    const size_t mats = std::max<size_t>(1, wpool_bytes / per);
    const size_t launches = std::max<size_t>(mats, 16);
    static bool warned = false;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (lds > 160 * 1024) { if (!warned) printf("  (skipping configs over 160 KiB LDS)\n"); warned = true; return; }
    auto one = [&](size_t i) {
        char* base = (char*)wpool + (i % mats) * per;
        P p{(const unsigned*)base, (const unsigned*)(base + wbytes + sbytes), (const f16*)(base + wbytes), xbuf, outbuf, M, s.K, s.N, R, 4};
        hipLaunchKernelGGL(kern, dim3(s.N / 16), dim3(W * 64), lds, st, p);
    };
    Res r; r.us = time_graph(one, launches, 5);
    snprintf(r.name, sizeof r.name, "%s", name);
    res.push_back(r);
}

template <typename KERN>
static void run_kernel3(std::vector<Res>& res, KERN kern, const Shape& s, int M, int W, int S, size_t lds, const char* name) {
    const int R = s.K / 8, G = s.K / 128;
    const size_t wbytes = (size_t)R * s.N * 4, cbytes = (size_t)(s.N / 16) * G * 48;
    const size_t per = wbytes + ((cbytes + 255) / 256) * 256;
    const size_t mats = std::max<size_t>(1, wpool_bytes / per);
    const size_t launches = std::max<size_t>(mats, 16);
    if (lds > 160 * 1024) return;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto one = [&](size_t i) {
        char* base = (char*)wpool + (i % mats) * per;
        P p{(const unsigned*)base, nullptr, nullptr, xbuf, outbuf, M, s.K, s.N, R, 4};
        hipLaunchKernelGGL(kern, dim3(s.N / 16 / S), dim3(W * 64), lds, st, p, (const unsigned char*)(base + wbytes));
    };
    Res r; r.us = time_graph(one, launches, 5);
    snprintf(r.name, sizeof r.name, "%s", name);
    res.push_back(r);
}
#define V3(MT, U, S, ABL, Wv) do { char nm[160]; snprintf(nm, sizeof nm, "V3 W=%2d U=%d S=%d abl=%2d", Wv, U, S, ABL); \
    const size_t lds = (size_t)MT * (s.K * 2 + 16) + (size_t)S * (s.K / 128) * 48 + (size_t)Wv * S * MT * 16 * 4 + 64; \
    if (Wv > 8) run_kernel3(res, dec_v3<MT, U, S, ABL, 16>, s, M, Wv, S, lds, nm); else run_kernel3(res, dec_v3<MT, U, S, ABL, 8>, s, M, Wv, S, lds, nm); } while (0)
#define V1(MT, U, DMA, ABL, Wv) do { char nm[160]; snprintf(nm, sizeof nm, "V1 W=%2d U=%d %s abl=%2d", Wv, U, DMA ? "dma" : "reg", ABL); \
    const size_t lds = (DMA ? (size_t)Wv * U * 1024 : 0) + (size_t)Wv * MT * 16 * 4 + 64; \
    if (Wv > 8) run_kernel(res, dec_v1<MT, U, DMA, ABL, 16>, s, M, Wv, lds, nm); else run_kernel(res, dec_v1<MT, U, DMA, ABL, 8>, s, M, Wv, lds, nm); } while (0)
#define V2(MT, U, DMA, XS, ABL, Wv) do { char nm[160]; snprintf(nm, sizeof nm, "V2 W=%2d U=%d %s x=%s abl=%2d", Wv, U, DMA ? "dma" : "reg", XS ? "lds" : "L2 ", ABL); \
    const size_t lds = (XS ? (size_t)MT * (s.K * 2 + 16) : 0) + (DMA ? (size_t)Wv * U * 1024 : 0) + (size_t)Wv * MT * 16 * 4 + 64; \
    if (Wv > 8) run_kernel(res, dec_v2<MT, U, DMA, XS, ABL, 16>, s, M, Wv, lds, nm); else run_kernel(res, dec_v2<MT, U, DMA, XS, ABL, 8>, s, M, Wv, lds, nm); } while (0)

__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
}
__global__ void fill_h(f16* p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (f16)(v * (1.f + 0.1f * (float)((i * 37) % 17) / 17.f));
}

int main(int argc, char** argv) {
    CK(hipStreamCreate(&st));
    wpool_bytes = (size_t)640 << 20;
    CK(hipMalloc(&wpool, wpool_bytes));
    CK(hipMalloc(&xbuf, 4 * 28672 * 2)); CK(hipMalloc(&outbuf, 4 * 32768 * 2));
    fill<<<2048, 256, 0, st>>>(wpool, wpool_bytes / 4);
    fill_h<<<64, 256, 0, st>>>(xbuf, 4 * 28672, 0.5f);
    CK(hipStreamSynchronize(st));
    const int M = 1;
    Shape shapes[] = {{4096, 4096, "o"}, {11008, 4096, "down"}, {4096, 12288, "q|k|v"}, {4096, 22016, "gate|up"}};
    for (auto s : shapes) {
        printf("== K=%d N=%d (%s), M=%d: %zu weight bytes per launch\n", s.K, s.N, s.what, M, (size_t)s.K / 8 * s.N * 4);
        std::vector<Res> res;
        const bool few = s.N <= 4096;
        if (few) {
            V1(1, 2, false, 0, 16); V2(1, 2, false, 1, 0, 16); V2(1, 4, false, 1, 0, 8);
            V3(1, 1, 1, 0, 16); V3(1, 2, 1, 0, 16); V3(1, 4, 1, 0, 16); V3(1, 2, 1, 0, 8); V3(1, 4, 1, 0, 8); V3(1, 8, 1, 0, 8); V3(1, 4, 1, 0, 4); V3(1, 8, 1, 0, 4);
            V3(1, 2, 1, 1, 16); V3(1, 2, 1, 2, 16); V3(1, 2, 1, 4, 16); V3(1, 2, 1, 8, 16); V3(1, 2, 1, 15, 16); V3(1, 2, 1, 7, 16);
            V3(1, 2, 2, 0, 16); V3(1, 4, 2, 0, 16); V3(1, 4, 2, 0, 8);
        } else {
            V1(1, 8, true, 0, 4); V1(1, 2, false, 0, 8); V2(1, 4, false, 1, 0, 4); V2(1, 8, false, 1, 0, 4);
            V3(1, 2, 1, 0, 4); V3(1, 4, 1, 0, 4); V3(1, 8, 1, 0, 4); V3(1, 4, 1, 0, 8); V3(1, 8, 1, 0, 2); V3(1, 4, 1, 0, 2);
            V3(1, 2, 2, 0, 4); V3(1, 4, 2, 0, 4); V3(1, 8, 2, 0, 4); V3(1, 4, 2, 0, 8); V3(1, 8, 2, 0, 8); V3(1, 4, 2, 0, 16); V3(1, 8, 2, 0, 2);
            V3(1, 4, 4, 0, 4); V3(1, 8, 4, 0, 4); V3(1, 4, 4, 0, 8); V3(1, 8, 4, 0, 8); V3(1, 4, 4, 0, 16); V3(1, 8, 4, 0, 16);
            V3(1, 4, 2, 1, 4); V3(1, 4, 2, 2, 4); V3(1, 4, 2, 4, 4); V3(1, 4, 2, 8, 4); V3(1, 4, 2, 15, 4); V3(1, 4, 2, 7, 4);
        }
        for (auto& r : res) printf("  %8.2f us  %7.1f GB/s  %s\n", r.us, (double)s.K / 8 * s.N * 4 / r.us / 1e3, r.name);
        fflush(stdout);
    }
    return 0;
}
