// gemvlab.hip -- ablation bench for the 4-bit fp16 decode GEMV structure (measurement tool, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I autogptq_amd/csrc tools/gemvlab.hip -o tools/gemvlab
// Each launch handles one [K/8, N] packed matrix out of a rotating > 256 MiB set (HBM-cold), M = 1.
// ABL bit 0: skip the x load; bit 1: skip scale/zero loads; bit 2: skip the math (xor the words);
// bit 3: skip the cross-lane / cross-wave reduction (lane 0 writes its own sum).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string.h>
#include "gemv.hip"   // the product kernels + plan/launch (so the lab can time them without torch / hipGraph)
using namespace gptq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
    const unsigned* qweight; const unsigned* qzeros; const f16* scales; const f16* x; f16* out; float* partial;
    int K, N, units_total, units_per_split, ksplit, gshift;
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int LN, int U, int ABL, bool MFMA>
__global__ void __launch_bounds__(1024) lab_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;
    constexpr int WR = 64 / LN, CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = strip * CT + cl * 4;
    const int ub = blockIdx.y * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    const int zrow_words = p.N >> 3;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rows_per_iter = W * WR * U;
    for (int base = ub; base < ue; base += rows_per_iter) {
        const int u0 = base + (wave * WR + rs) * U;
        u32x4 q[U], xr[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int ul = min(u0 + j, ue - 1);
            q[j] = __builtin_nontemporal_load((const u32x4*)(p.qweight + (size_t)ul * p.N + n0));
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if constexpr (ABL & 1) xr[j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
            else xr[j] = *(const u32x4*)(p.x + (size_t)min(u0 + j, ue - 1) * 8);
        }
        const int g = min(u0, ue - 1) >> p.gshift;
        u32x2 sraw = {0x14001400u, 0x14001400u};
        unsigned zw = 0x7777u;
        if constexpr (!(ABL & 2)) {
            sraw = *(const u32x2*)(p.scales + (size_t)g * p.N + n0);
            zw = p.qzeros[(size_t)g * zrow_words + (n0 >> 3)] >> ((n0 & 7) * 4);
        }
        if constexpr (ABL & 4) {
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c][0] += as_f32((q[j][c] ^ xr[j][c] ^ sraw[c & 1] ^ zw) & 0x3fffffffu);
            continue;
        }
        f16x2 c1[4], c2[4];
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned z = (((zw >> (4 * c)) & 15u) + 1u) & 15u;
            c1[c] = as_f16x2(z * 0x00010001u + 0xE400E400u);
            c2[c] = c1[c] + k960;
        }
        f32x4 accg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32x4 qv = q[j];
            const u32x4 t = xr[j];
            const bool live = (u0 + j < ue);
            u32x2 a01 = {__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u)};
            u32x2 a23 = {__builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
            if (!live) { a01 = u32x2{0u, 0u}; a23 = u32x2{0u, 0u}; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned qw = qv[c], q8 = qw >> 8;
                const f16x2 h0 = as_f16x2((qw & 0x000f000fu) | 0x64006400u) + c1[c];
                const f16x2 h1 = as_f16x2((qw & 0x00f000f0u) | 0x64006400u) * r16 + c2[c];
                const f16x2 h2 = as_f16x2((q8 & 0x000f000fu) | 0x64006400u) + c1[c];
                const f16x2 h3 = as_f16x2((q8 & 0x00f000f0u) | 0x64006400u) * r16 + c2[c];
                if constexpr (MFMA) {
                    const u32x2 b01 = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                    const u32x2 b23 = {__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
                    accg[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a01), __builtin_bit_cast(f16x4, b01), accg[c], 0, 0, 0);
                    accg[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a23), __builtin_bit_cast(f16x4, b23), accg[c], 0, 0, 0);
                } else {
                    float d = __builtin_amdgcn_fdot2(h0, as_f16x2(a01[0]), 0.f, false);
                    d = __builtin_amdgcn_fdot2(h1, as_f16x2(a01[1]), d, false);
                    d = __builtin_amdgcn_fdot2(h2, as_f16x2(a23[0]), d, false);
                    d = __builtin_amdgcn_fdot2(h3, as_f16x2(a23[1]), d, false);
                    accg[c][0] += d;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned sh = (c & 1) ? (sraw[c >> 1] >> 16) : (sraw[c >> 1] & 0xffffu);
            const float sc = (float)as_f16((unsigned short)sh);
            acc[c][0] = fmaf(sc, accg[c][0], acc[c][0]);
        }
    }
    if constexpr (ABL & 8) {
        if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 1.2345f) p.out[n0] = (f16)acc[0][0];
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = acc[c][0];
#pragma unroll
        for (int off = LN; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        acc[c][0] = v;
    }
    if (lane < LN) {
        f32x4 v = {acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
        *(f32x4*)(red + wave * CT + lane * 4) = v;
    }
    __syncthreads();
    for (int i = tid; i < CT; i += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += red[w * CT + i];
        const int n = strip * CT + i;
        if (p.ksplit > 1) p.partial[(size_t)blockIdx.y * p.N + n] = s;
        else p.out[n] = (f16)s;
    }
}


// ---- v2: small (L2-hit) loads first, software-pipelined chunks, DPP reduce ---------------------
template <int U> struct Chunk { u32x4 q[U], xr[U]; u32x2 sraw; unsigned zw; int u0; };

template <int LN, int U, int DEPTH, bool MFMA>
__global__ void __launch_bounds__(1024) lab2_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;
    constexpr int WR = 64 / LN, CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = strip * CT + cl * 4;
    const int ub = blockIdx.y * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    const int zrow_words = p.N >> 3;
    const unsigned* __restrict__ qcol = p.qweight + n0;
    const f16* __restrict__ scol = p.scales + n0;
    const unsigned* __restrict__ zcol = p.qzeros + (n0 >> 3);
    const int zsh = (n0 & 7) * 4;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rows_per_iter = W * WR * U;
    const int lane_row = (wave * WR + rs) * U;

    auto issue = [&](Chunk<U>& c, int base) {
        c.u0 = base + lane_row;
        const int g = min(c.u0, ue - 1) >> p.gshift;
        c.sraw = *(const u32x2*)(scol + (size_t)g * p.N);
        c.zw = zcol[(size_t)g * zrow_words] >> zsh;
#pragma unroll
        for (int j = 0; j < U; ++j) c.xr[j] = *(const u32x4*)(p.x + (size_t)min(c.u0 + j, ue - 1) * 8);
#pragma unroll
        for (int j = 0; j < U; ++j) c.q[j] = __builtin_nontemporal_load((const u32x4*)(qcol + (size_t)min(c.u0 + j, ue - 1) * p.N));
    };
    auto compute = [&](const Chunk<U>& ch) {
        f16x2 c1[4], c2[4];
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned z = (((ch.zw >> (4 * c)) & 15u) + 1u) & 15u;
            c1[c] = as_f16x2(z * 0x00010001u + 0xE400E400u);
            c2[c] = c1[c] + k960;
        }
        f32x4 accg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32x4 qv = ch.q[j];
            const u32x4 t = ch.xr[j];
            const bool live = (ch.u0 + j < ue);
            u32x2 a01 = {__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u)};
            u32x2 a23 = {__builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
            if (!live) { a01 = u32x2{0u, 0u}; a23 = u32x2{0u, 0u}; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned qw = qv[c], q8 = qw >> 8;
                const f16x2 h0 = as_f16x2((qw & 0x000f000fu) | 0x64006400u) + c1[c];
                const f16x2 h1 = as_f16x2((qw & 0x00f000f0u) | 0x64006400u) * r16 + c2[c];
                const f16x2 h2 = as_f16x2((q8 & 0x000f000fu) | 0x64006400u) + c1[c];
                const f16x2 h3 = as_f16x2((q8 & 0x00f000f0u) | 0x64006400u) * r16 + c2[c];
                if constexpr (MFMA) {
                    const u32x2 b01 = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                    const u32x2 b23 = {__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
                    accg[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a01), __builtin_bit_cast(f16x4, b01), accg[c], 0, 0, 0);
                    accg[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a23), __builtin_bit_cast(f16x4, b23), accg[c], 0, 0, 0);
                } else {
                    float d = __builtin_amdgcn_fdot2(h0, as_f16x2(a01[0]), 0.f, false);
                    d = __builtin_amdgcn_fdot2(h1, as_f16x2(a01[1]), d, false);
                    d = __builtin_amdgcn_fdot2(h2, as_f16x2(a23[0]), d, false);
                    d = __builtin_amdgcn_fdot2(h3, as_f16x2(a23[1]), d, false);
                    accg[c][0] += d;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned sh = (c & 1) ? (ch.sraw[c >> 1] >> 16) : (ch.sraw[c >> 1] & 0xffffu);
            const float sc = (float)as_f16((unsigned short)sh);
            acc[c][0] = fmaf(sc, accg[c][0], acc[c][0]);
        }
    };

    if constexpr (DEPTH == 1) {
        Chunk<U> a;
        for (int base = ub; base < ue; base += rows_per_iter) { issue(a, base); compute(a); }
    } else {
        Chunk<U> a, b;
        issue(a, ub);
        for (int base = ub; base < ue; base += 2 * rows_per_iter) {
            issue(b, base + rows_per_iter);          // clamped: past-the-end chunks re-read the last row, contribute 0
            compute(a);
            issue(a, base + 2 * rows_per_iter);
            compute(b);
        }
    }
    // reduce over row slots: DPP rotates inside a 16-lane row, bpermute across rows
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = acc[c][0];
        if constexpr (LN <= 4) v = dpp_add<0x124>(v);    // row_ror:4
        if constexpr (LN <= 8) v = dpp_add<0x128>(v);    // row_ror:8
        if constexpr (LN <= 16) v += __shfl_xor(v, 16, 64);
        if constexpr (LN <= 32) v += __shfl_xor(v, 32, 64);
        acc[c][0] = v;
    }
    if (lane < LN) {
        f32x4 v = {acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
        *(f32x4*)(red + wave * CT + lane * 4) = v;
    }
    __syncthreads();
    for (int i = tid; i < CT; i += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += red[w * CT + i];
        const int n = strip * CT + i;
        if (p.ksplit > 1) p.partial[(size_t)blockIdx.y * p.N + n] = s;
        else p.out[n] = (f16)s;
    }
}

__global__ void __launch_bounds__(256) reduce_kernel(const float* __restrict__ partial, f16* __restrict__ out, int S, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += partial[(size_t)k * N + i];
    out[i] = (f16)s;
}

__global__ void fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7) ^ seed;
        v ^= v << 13; v ^= v >> 17; v ^= v << 5;
        p[i] = v;
    }
}
__global__ void fill_f16(f16* p, size_t n, float lo, float hi) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)(i * 2246822519u) ^ 0x9e3779b9u; v ^= v >> 15; v *= 2654435761u; v ^= v >> 13;
        p[i] = (f16)(lo + (hi - lo) * (float)(v & 0xffff) / 65536.f);
    }
}

struct Bufs { unsigned* qw; unsigned* qz; f16* sc; f16* x; f16* out; float* partial; size_t mats; };

template <int LN, int U, int ABL, bool MFMA>
float run(const Bufs& b, int K, int N, int waves, int ksplit, int reps, hipStream_t st) {
    P p{};
    p.K = K; p.N = N; p.units_total = K / 8; p.ksplit = ksplit;
    p.units_per_split = (p.units_total + ksplit - 1) / ksplit;
    p.gshift = 4;
    p.x = b.x; p.out = b.out; p.partial = b.partial;
    const int strips = N / (LN * 4);
    dim3 grid(strips, ksplit), block(waves * 64);
    const size_t lds = (size_t)waves * LN * 4 * 4;
    const size_t qw_words = (size_t)K / 8 * N, qz_words = (size_t)(K / 128) * N / 8, sc_elems = (size_t)(K / 128) * N;
    auto launch_all = [&]() {
        for (size_t i = 0; i < b.mats; ++i) {
            p.qweight = b.qw + i * qw_words; p.qzeros = b.qz + i * qz_words; p.scales = b.sc + i * sc_elems;
            lab_kernel<LN, U, ABL, MFMA><<<grid, block, lds, st>>>(p);
            if (ksplit > 1 && !(ABL & 8)) reduce_kernel<<<(N + 255) / 256, 256, 0, st>>>(b.partial, b.out, ksplit, N);
        }
    };
    launch_all();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) launch_all();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * b.mats);
}


template <int LN, int U, int DEPTH, bool MFMA>
float run2(const Bufs& b, int K, int N, int waves, int ksplit, int reps, hipStream_t st, bool check = false) {
    P p{};
    p.K = K; p.N = N; p.units_total = K / 8; p.ksplit = ksplit;
    p.units_per_split = (p.units_total + ksplit - 1) / ksplit;
    p.gshift = 4;
    p.x = b.x; p.out = b.out; p.partial = b.partial;
    const int strips = N / (LN * 4);
    dim3 grid(strips, ksplit), block(waves * 64);
    const size_t lds = (size_t)waves * LN * 4 * 4;
    const size_t qw_words = (size_t)K / 8 * N, qz_words = (size_t)(K / 128) * N / 8, sc_elems = (size_t)(K / 128) * N;
    auto launch_all = [&]() {
        for (size_t i = 0; i < b.mats; ++i) {
            p.qweight = b.qw + i * qw_words; p.qzeros = b.qz + i * qz_words; p.scales = b.sc + i * sc_elems;
            lab2_kernel<LN, U, DEPTH, MFMA><<<grid, block, lds, st>>>(p);
            if (ksplit > 1) reduce_kernel<<<(N + 255) / 256, 256, 0, st>>>(b.partial, b.out, ksplit, N);
        }
    };
    launch_all();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) launch_all();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * b.mats);
}

// last matrix's output against the plain lab_kernel<4,2,0,false> result (same math, different order)
static double checksum(const Bufs& b, int N, hipStream_t st) {
    std::vector<unsigned short> h(N);
    CK(hipMemcpyAsync(h.data(), b.out, (size_t)N * 2, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double s = 0;
    for (int i = 0; i < N; ++i) { _Float16 v; memcpy(&v, &h[i], 2); s += (double)(float)v * (1 + (i % 7)); }
    return s;
}

// time the PRODUCT path (plan_gemv + launch_gemv) in this harness
float run_product(const Bufs& b, int K, int N, int M, const gptq_tuning_t* tune, int reps, hipStream_t st, char* desc, size_t dn) {
    const size_t qw_words = (size_t)K / 8 * N, qz_words = (size_t)(K / 128) * N / 8, sc_elems = (size_t)(K / 128) * N;
    gptq_layer_t L{};
    L.K = K; L.N = N; L.bits = 4; L.group_size = 128; L.dtype = GPTQ_F16; L.zero_mode = GPTQ_ZERO_WRAP;
    L.qweight = b.qw; L.qzeros = b.qz; L.scales = b.sc;
    GemvPlan pl = plan_gemv(L, M, tune);
    snprintf(desc, dn, "product M=%d: %s ln=%d waves=%d ksplit=%d u=%d mt=%d", M, pl.mfma ? "mfma" : pl.direct ? "direct" : pl.fast ? "lds" : "generic",
             pl.ln, pl.waves, pl.ksplit, pl.u, pl.mt);
    auto launch_all = [&]() {
        for (size_t i = 0; i < b.mats; ++i) {
            L.qweight = b.qw + i * qw_words; L.qzeros = b.qz + i * qz_words; L.scales = b.sc + i * sc_elems;
            hipError_t e = launch_gemv(L, pl, b.x, b.out, M, b.partial, st);
            if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
        }
    };
    launch_all();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) launch_all();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * b.mats);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shape { int K, N; } shapes[] = {{4096, 4096}, {4096, 11008}, {11008, 4096}};
    for (auto s : shapes) {
        const int K = s.K, N = s.N;
        const size_t qw_b = (size_t)K / 8 * N * 4, qz_b = (size_t)(K / 128) * N / 8 * 4, sc_b = (size_t)(K / 128) * N * 2;
        Bufs b{};
        b.mats = ((size_t)640 << 20) / qw_b;
        CK(hipMalloc(&b.qw, qw_b * b.mats)); CK(hipMalloc(&b.qz, qz_b * b.mats)); CK(hipMalloc(&b.sc, sc_b * b.mats));
        CK(hipMalloc(&b.x, (size_t)K * 2 * 8)); CK(hipMalloc(&b.out, (size_t)N * 2 * 8)); CK(hipMalloc(&b.partial, (size_t)64 * N * 4 * 8));
        fill<<<2048, 256, 0, st>>>(b.qw, qw_b * b.mats / 4, 1u);
        fill<<<256, 256, 0, st>>>(b.qz, qz_b * b.mats / 4, 2u);
        fill_f16<<<256, 256, 0, st>>>(b.sc, sc_b * b.mats / 2, 0.002f, 0.0022f);
        fill_f16<<<64, 256, 0, st>>>(b.x, (size_t)K * 8, -0.5f, 0.5f);
        CK(hipStreamSynchronize(st));
        printf("== K=%d N=%d : %zu B packed per launch, %zu rotating matrices\n", K, N, qw_b, b.mats);
        struct R { float us; char name[200]; };
        std::vector<R> res;
#define TRY(LN, U, ABL, MF, W, KS) do { if (N % (LN * 4) == 0) { R r; r.us = run<LN, U, ABL, MF>(b, K, N, W, KS, 3, st); \
        snprintf(r.name, sizeof r.name, "LN=%2d U=%d waves=%2d ksplit=%2d %s abl=%2d [%s%s%s%s]", LN, U, W, KS, MF ? "mfma" : "dot2", ABL, \
                 (ABL & 1) ? "-x " : "", (ABL & 2) ? "-sz " : "", (ABL & 4) ? "-math " : "", (ABL & 8) ? "-reduce" : ""); res.push_back(r); } } while (0)
#define ABLS(LN, U, W, KS) TRY(LN, U, 0, true, W, KS); TRY(LN, U, 0, false, W, KS); TRY(LN, U, 1, true, W, KS); TRY(LN, U, 2, true, W, KS); \
        TRY(LN, U, 3, true, W, KS); TRY(LN, U, 4, true, W, KS); TRY(LN, U, 7, true, W, KS); TRY(LN, U, 8, true, W, KS); TRY(LN, U, 15, true, W, KS); TRY(LN, U, 11, true, W, KS)
        if (K == 4096 && N == 4096) { ABLS(4, 2, 16, 1); }
        else if (N == 11008) { ABLS(4, 2, 16, 1); }
        else { ABLS(4, 4, 16, 1); }
#define TRY2(LN, U, D, MF, W, KS) do { if (N % (LN * 4) == 0) { R r; r.us = run2<LN, U, D, MF>(b, K, N, W, KS, 3, st); \
        snprintf(r.name, sizeof r.name, "v2 LN=%2d U=%d depth=%d waves=%2d ksplit=%2d %s  checksum %.4f", LN, U, D, W, KS, MF ? "mfma" : "dot2", checksum(b, N, st)); res.push_back(r); } } while (0)
#define V2S(LN, U, W, KS) TRY2(LN, U, 1, true, W, KS); TRY2(LN, U, 2, true, W, KS); TRY2(LN, U, 1, false, W, KS); TRY2(LN, U, 2, false, W, KS)
        {
            struct T { int path, ln, waves, ks, u, M; } ts[] = {{0,0,0,0,0,1},{5,4,16,1,2,1},{5,4,16,1,1,1},{5,4,8,1,2,1},{5,8,16,1,2,1},{4,4,16,1,2,1},{2,4,16,1,0,1},{2,0,0,0,0,1},
                                                          {0,0,0,0,0,2},{0,0,0,0,0,4},{0,0,0,0,0,8},{1,0,0,0,0,1}};
            for (auto t : ts) {
                gptq_tuning_t tu{}; tu.path = t.path; tu.lanes_n = t.ln; tu.waves = t.waves; tu.ksplit = t.ks; tu.reserved[0] = t.u;
                R r; char d[100];
                r.us = run_product(b, K, N, t.M, &tu, 3, st, d, sizeof d);
                snprintf(r.name, sizeof r.name, "%s  checksum %.4f", d, checksum(b, N, st)); res.push_back(r);
            }
        }
        { R r; r.us = run<4, 2, 0, false>(b, K, N, 16, 1, 1, st); snprintf(r.name, sizeof r.name, "reference checksum %.4f", checksum(b, N, st)); res.push_back(r); }
        V2S(4, 1, 16, 1); V2S(4, 2, 16, 1); V2S(4, 2, 8, 1);
        for (size_t i = 0; i < res.size(); ++i) printf("  %8.2f us  %7.1f GB/s  %s\n", res[i].us, qw_b / res[i].us / 1e3, res[i].name);
        CK(hipFree(b.qw)); CK(hipFree(b.qz)); CK(hipFree(b.sc)); CK(hipFree(b.x)); CK(hipFree(b.out)); CK(hipFree(b.partial));
    }
    return 0;
}
