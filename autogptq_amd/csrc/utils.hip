// utils.hip -- layout kernels either side of the matmul: integer unpack, full dequant
// ("reconstruct"), device pack(), act-order row re-sequencing, x column permutation.
//
// Reference behaviour being reproduced (AutoGPTQ v0.8.0.dev0; nothing is copied):
//   unpack / dequant   auto_gptq/nn_modules/qlinear/qlinear_cuda_old.py:295-349, qlinear_cuda.py:257-302
//   pack               auto_gptq/nn_modules/qlinear/qlinear_cuda.py:108-203
//   re-sequencing      semantics of Q4Matrix::make_sequential, autogptq_extension/exllama/cuda_func/q4_matrix.cu:63-169
//   column remap       autogptq_extension/exllama/cuda_func/column_remap.cu:9-63
// All of these are HBM-bound integer/byte kernels: lanes run along N (the contiguous axis of
// every GPTQ tensor) with 4 columns (16 B of packed words) per lane.
#include "common.cuh"
#include "launch.h"

namespace gptq {

// ---------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(256) unpack_weights_kernel(const unsigned* __restrict__ qweight, int units, int N,
                                                             uint8_t* __restrict__ out) {
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals;
    const int n0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int u = blockIdx.y;
    if (n0 >= N || u >= units) return;
    u32x4 q[UW];
#pragma unroll
    for (int w = 0; w < UW; ++w) q[w] = *(const u32x4*)(qweight + (size_t)(u * UW + w) * N + n0);
    [&]<int... V>(std::integer_sequence<int, V...>) {
        (([&] {
             unsigned o = 0;
#pragma unroll
             for (int c = 0; c < 4; ++c) {
                 unsigned w[UW];
#pragma unroll
                 for (int i = 0; i < UW; ++i) w[i] = q[i][c];
                 o |= unit_field<BITS, V>(w) << (8 * c);
             }
             *(unsigned*)(out + (size_t)(u * KPU + V) * N + n0) = o;
         }()),
         ...);
    }(std::make_integer_sequence<int, KPU>{});
}

__global__ void __launch_bounds__(256) unpack_zeros_kernel(const unsigned* __restrict__ qzeros, int G, int N, int bits,
                                                           int zero_mode, int* __restrict__ out) {
    const int n0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int g = blockIdx.y;
    if (n0 >= N || g >= G) return;
    int z[4];
    zero_points4(qzeros + (size_t)g * (N / 32 * bits), n0, bits, zero_mode, z);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[(size_t)g * N + n0 + c] = z[c];
}

// W[k,n] = T( float(scale[g,n]) * float(w - z) ): the fp32 product is exact for 16-bit scales, so
// this is the correctly rounded product = what torch computes for scales * (weight - zeros).
template <int BITS, typename T>
__global__ void __launch_bounds__(256) dequant_kernel(const unsigned* __restrict__ qweight, const unsigned* __restrict__ qzeros,
                                                      const T* __restrict__ scales, const int* __restrict__ g_idx,
                                                      int units, int N, int group_size, int zero_mode,
                                                      T* __restrict__ out) {
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals;
    const int n0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int u = blockIdx.y;
    if (n0 >= N || u >= units) return;
    u32x4 q[UW];
#pragma unroll
    for (int w = 0; w < UW; ++w) q[w] = *(const u32x4*)(qweight + (size_t)(u * UW + w) * N + n0);
    const int zrow_words = N / 32 * BITS;
    [&]<int... V>(std::integer_sequence<int, V...>) {
        (([&] {
             const int k = u * KPU + V;
             const int g = g_idx ? g_idx[k] : k / group_size;
             int z[4];
             zero_points4(qzeros + (size_t)g * zrow_words, n0, BITS, zero_mode, z);
#pragma unroll
             for (int c = 0; c < 4; ++c) {
                 unsigned w[UW];
#pragma unroll
                 for (int i = 0; i < UW; ++i) w[i] = q[i][c];
                 const float s = DType<T>::to_f32(scales[(size_t)g * N + n0 + c]);
                 out[(size_t)k * N + n0 + c] = DType<T>::from_f32(s * (float)((int)unit_field<BITS, V>(w) - z[c]));
             }
         }()),
         ...);
    }(std::make_integer_sequence<int, KPU>{});
}

// ---------------------------------------------------------------------------------------------
// pack(): arithmetic in torch's promoted dtype, rounding after every op like the CPU reference.
__device__ __forceinline__ float load_as_f32(const void* p, size_t i, int dt) {
    switch (dt) {
        case GPTQ_F16: return (float)((const f16*)p)[i];
        case GPTQ_BF16: return (float)((const bf16*)p)[i];
        default: return ((const float*)p)[i];
    }
}
__device__ __forceinline__ float round_to(float v, int dt) {
    switch (dt) {
        case GPTQ_F16: return (float)(f16)v;
        case GPTQ_BF16: return (float)(bf16)v;
        default: return v;
    }
}
__host__ __device__ inline int promote(int a, int b) { return (a == b) ? a : GPTQ_F32; }

// W is [N, K] (rows = output features): a workgroup stages a 32-column x 256-k tile through LDS with loads that run along K
// (512 contiguous bytes per W row for fp16), then every thread quantises and packs whole units for consecutive columns n --
// so both the W reads and the qweight / scales accesses are coalesced (the first version read W[n, k] with lanes along n,
// one 16-64 byte piece per lane at stride K).
template <int BITS>
__global__ void __launch_bounds__(256) pack_weights_kernel(const void* __restrict__ W, const void* __restrict__ scale_in,
                                                           const void* __restrict__ zero_in, const int* __restrict__ g_idx,
                                                           int K, int N, int group_size, int w_dt, int q_dt,
                                                           unsigned* __restrict__ qweight, void* __restrict__ scales_out) {
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals, TN = 32, TK = 256;
    __shared__ float tile[TN][TK + 1];
    const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
    const int tid = threadIdx.x;
    for (int r = 0; r < TN; ++r) {
        const int n = n0 + r, k = k0 + tid;
        tile[r][tid] = (n < N && k < K) ? load_as_f32(W, (size_t)n * K + k, w_dt) : 0.f;
    }
    __syncthreads();
    const int p_dt = promote(w_dt, q_dt);
    for (int idx = tid; idx < TN * (TK / KPU); idx += 256) {
        const int ul = idx / TN, nl = idx - ul * TN;
        const int n = n0 + nl, u = k0 / KPU + ul;
        if (n >= N || u >= K / KPU) continue;
        unsigned vals[KPU];
#pragma unroll
        for (int v = 0; v < KPU; ++v) {
            const int k = u * KPU + v;
            const int g = g_idx ? g_idx[k] : k / group_size;
            const float s_in = load_as_f32(scale_in, (size_t)g * N + n, q_dt);
            const float z_in = load_as_f32(zero_in, (size_t)g * N + n, q_dt);
            const float sz = round_to(z_in * s_in, q_dt);                     // scale_zeros = zeros * scales
            const float s_cast = round_to(s_in, w_dt);                         // self.scales (layer dtype)
            const float w = tile[nl][ul * KPU + v];
            const float sum = round_to(w + sz, p_dt);
            const float quo = round_to(sum / s_cast, p_dt);
            vals[v] = (unsigned)(int)rintf(quo);                               // torch.round = half-to-even
        }
        unsigned w[UW];
#pragma unroll
        for (int i = 0; i < UW; ++i) w[i] = 0u;
        if constexpr (BITS != 3) {
#pragma unroll
            for (int v = 0; v < KPU; ++v) w[0] |= vals[v] << (BITS * v);        // unmasked OR, like the reference
        } else {
#pragma unroll
            for (int j = 0; j < 10; ++j) w[0] |= vals[j] << (3 * j);
            w[0] |= vals[10] << 30;
            w[1] |= (vals[10] >> 2) & 1u;
#pragma unroll
            for (int j = 0; j < 10; ++j) w[1] |= vals[11 + j] << (3 * j + 1);
            w[1] |= vals[21] << 31;
            w[2] |= (vals[21] >> 1) & 3u;
#pragma unroll
            for (int j = 0; j < 10; ++j) w[2] |= vals[22 + j] << (3 * j + 2);
        }
#pragma unroll
        for (int i = 0; i < UW; ++i) qweight[(size_t)(u * UW + i) * N + n] = w[i];
    }
    // scales_out = scales.to(layer dtype): the workgroups of the first K tile write their 32 columns of every group row
    if (scales_out && blockIdx.y == 0) {
        const int G = (K + group_size - 1) / group_size;
        for (int idx = tid; idx < G * TN; idx += 256) {
            const int g = idx / TN, n = n0 + (idx - g * TN);
            if (n >= N) continue;
            const float s = round_to(load_as_f32(scale_in, (size_t)g * N + n, q_dt), w_dt);
            switch (w_dt) {
                case GPTQ_F16: ((f16*)scales_out)[(size_t)g * N + n] = (f16)s; break;
                case GPTQ_BF16: ((bf16*)scales_out)[(size_t)g * N + n] = (bf16)s; break;
                default: ((float*)scales_out)[(size_t)g * N + n] = s;
            }
        }
    }
}

template <int BITS>
__global__ void __launch_bounds__(256) pack_zeros_kernel(const void* __restrict__ zero_in, int G, int N, int q_dt,
                                                         unsigned* __restrict__ qzeros) {
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals;
    const int cu = blockIdx.x * blockDim.x + threadIdx.x;   // column unit
    const int g = blockIdx.y;
    if (cu >= N / KPU || g >= G) return;
    unsigned vals[KPU];
#pragma unroll
    for (int v = 0; v < KPU; ++v) {
        const float z = round_to(load_as_f32(zero_in, (size_t)g * N + cu * KPU + v, q_dt) - 1.0f, q_dt);  // zeros -= 1
        vals[v] = (unsigned)(long long)z;                    // numpy .astype(uint32): -1.0 -> 0xFFFFFFFF
    }
    unsigned w[UW];
#pragma unroll
    for (int i = 0; i < UW; ++i) w[i] = 0u;
    if constexpr (BITS != 3) {
#pragma unroll
        for (int v = 0; v < KPU; ++v) w[0] |= vals[v] << (BITS * v);
    } else {
#pragma unroll
        for (int j = 0; j < 10; ++j) w[0] |= vals[j] << (3 * j);
        w[0] |= vals[10] << 30;
        w[1] |= (vals[10] >> 2) & 1u;
#pragma unroll
        for (int j = 0; j < 10; ++j) w[1] |= vals[11 + j] << (3 * j + 1);
        w[1] |= vals[21] << 31;
        w[2] |= (vals[21] >> 1) & 3u;
#pragma unroll
        for (int j = 0; j < 10; ++j) w[2] |= vals[22 + j] << (3 * j + 2);
    }
#pragma unroll
    for (int i = 0; i < UW; ++i) qzeros[(size_t)g * (N / 32 * BITS) + cu * UW + i] = w[i];
}

// ---------------------------------------------------------------------------------------------
// out row-stream position i holds the field of source k = perm[i] (masked fields, clean re-pack).
template <int BITS>
__global__ void __launch_bounds__(256) resequence_kernel(const unsigned* __restrict__ qweight, const int* __restrict__ perm,
                                                         int units, int N, unsigned* __restrict__ out) {
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int u = blockIdx.y;
    if (n >= N || u >= units) return;
    unsigned long long lo = 0ull;  // bits 0..63 of the unit stream
    unsigned hi = 0u;              // bits 64..95 (3-bit only)
#pragma unroll
    for (int v = 0; v < KPU; ++v) {
        const unsigned f = stream_field(qweight + n, (size_t)N, (unsigned)perm[u * KPU + v], BITS);
        const int bit = BITS * v;
        if (bit < 64) {
            lo |= (unsigned long long)f << bit;
            if (bit + BITS > 64) hi |= f >> (64 - bit);
        } else {
            hi |= f << (bit - 64);
        }
    }
    out[(size_t)(u * UW) * N + n] = (unsigned)lo;
    if constexpr (UW == 3) {
        out[(size_t)(u * UW + 1) * N + n] = (unsigned)(lo >> 32);
        out[(size_t)(u * UW + 2) * N + n] = hi;
    }
}

// ---- the decode copy of a 2/3/4/8-bit layer (layouts: include/gptq_mi355x.h, gptq_layer_t.qweight_tiled / qconst_tiled; consumer: gemv_tiled.hip) --------
// Load time only.  One workgroup = one chunk (4 k-slots x 16 columns, a lane's WPL words adjacent); one thread = one stored word, built field by field
// from the checkpoint's bit stream of its column (stream_field: the 3-bit straddlers are resolved HERE, the decode kernel never sees one).
//   4-bit  stored nibble p of word w = k 8w + {0, 2, 4, 6, 1, 3, 5, 7}[p]: (q & 0x000f000f) then picks (k0, k1), (q & 0x00f000f0) (k2, k3), and the
//          same on q >> 8 (k4, k5), (k6, k7) -- the order x lies in memory
//   8-bit  stored byte p of word w = k 4w + {0, 2, 1, 3}[p]: (q & 0x00ff00ff) = (k0, k1), the same on q >> 8 = (k2, k3)
//   3-bit  word j of the lane's three: pair 5j + i (i = 0..4) = (k 2p, k 2p + 1) at bit 3i of the low / high 16 bits; bit 15 / 31 = bit j of k30 / k31
//   2-bit  (round 6) a lane holds TWO words = 32 k: word w holds pair p (p = 0..7) = (k 16w + 2p, k 16w + 2p + 1) at bit 2p of its low / high 16 bits
template <int BITS>
__global__ void __launch_bounds__(256) prepack_decode_weights_kernel(const unsigned* __restrict__ q, int K, int N, int chunks, unsigned* __restrict__ out) {
    constexpr int WPL = BITS == 3 ? 3 : (BITS == 2 ? 2 : 4), KPL = BITS == 8 ? 16 : 32, CKE = 4 * KPL;
    const int c = blockIdx.x, s = blockIdx.y, t = threadIdx.x;
    const int col = t & 15, w = (t >> 4) % WPL, kb = (t >> 4) / WPL;              // reads of one k run over 16 adjacent columns
    if (kb >= 4) return;
    const int n = s * 16 + col, k0 = c * CKE + kb * KPL;
    auto val = [&](int k) -> unsigned {                                          // the checkpoint's value (k, n); rows past K read as 0
        if (k >= K) return 0u;
        const int unit = k / Pack<BITS>::vals;
        return stream_field(q + (size_t)unit * Pack<BITS>::words * N + n, (size_t)N, (unsigned)(k - unit * Pack<BITS>::vals), BITS);
    };
    unsigned v = 0;
    if constexpr (BITS == 4) {
        constexpr int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
#pragma unroll
        for (int p = 0; p < 8; ++p) v |= val(k0 + 8 * w + order[p]) << (4 * p);
    } else if constexpr (BITS == 8) {
        constexpr int order[4] = {0, 2, 1, 3};
#pragma unroll
        for (int p = 0; p < 4; ++p) v |= val(k0 + 4 * w + order[p]) << (8 * p);
    } else if constexpr (BITS == 2) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            v |= val(k0 + 16 * w + 2 * p) << (2 * p);
            v |= val(k0 + 16 * w + 2 * p + 1) << (16 + 2 * p);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pr = 5 * w + i;
            v |= val(k0 + 2 * pr) << (3 * i);
            v |= val(k0 + 2 * pr + 1) << (16 + 3 * i);
        }
        v |= ((val(k0 + 30) >> w) & 1u) << 15;
        v |= ((val(k0 + 31) >> w) & 1u) << 31;
    }
    out[((size_t)s * chunks + c) * (64 * WPL) + (kb * 16 + col) * WPL + w] = v;
}
// The inverse of the weights half (round 5: gptq_unprepack_decode): one thread = one word of the checkpoint layout [K/32*bits, N], rebuilt value by value from
// the copy -- what lets a layer whose checkpoint rows have left the HBM (QuantLinear.post_init(release_checkpoint_layout=True)) still serve the kernels that
// read rows, and what state_dict() / a re-save would be rebuilt from.  Reference behaviour it answers: the in-place re-layouts of exllama / exllamav2
// (q4_matrix.cu:160, q_matrix.cu:149) keep ONE copy of the weights on the device.
template <int BITS>
__global__ void __launch_bounds__(256) unprepack_decode_weights_kernel(const unsigned* __restrict__ t, int K, int N, int chunks, unsigned* __restrict__ out) {
    constexpr int WPL = BITS == 3 ? 3 : (BITS == 2 ? 2 : 4), KPL = BITS == 8 ? 16 : 32, CKE = 4 * KPL;
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;          // output word (row r, column n)
    if (n >= N) return;
    const int s = n >> 4, col = n & 15;
    auto val = [&](int k) -> unsigned {                                          // the copy's value (k, n)
        const int c = k / CKE, kb = (k - c * CKE) / KPL, j = k - c * CKE - kb * KPL;
        const unsigned* lw = t + ((size_t)s * chunks + c) * (64 * WPL) + (kb * 16 + col) * WPL;
        if constexpr (BITS == 4) {
            const int i = j & 7, p = (i & 1) ? 4 + (i >> 1) : (i >> 1);          // stored nibble order k0 k2 k4 k6 k1 k3 k5 k7
            return (lw[j >> 3] >> (4 * p)) & 15u;
        } else if constexpr (BITS == 8) {
            const int i = j & 3, p = i == 1 ? 2 : (i == 2 ? 1 : i);              // stored byte order k0 k2 k1 k3
            return (lw[j >> 2] >> (8 * p)) & 255u;
        } else if constexpr (BITS == 2) {
            const int i = j & 15;                                                // pair i >> 1 at bit 2 (i >> 1) of the low (even k) / high (odd k) half
            return (lw[j >> 4] >> (2 * (i >> 1) + 16 * (i & 1))) & 3u;
        } else {
            const int pr = j >> 1, hi = (j & 1) * 16;
            if (pr < 15) return (lw[pr / 5] >> (3 * (pr % 5) + hi)) & 7u;
            return ((lw[0] >> (15 + hi)) & 1u) | (((lw[1] >> (15 + hi)) & 1u) << 1) | (((lw[2] >> (15 + hi)) & 1u) << 2);
        }
    };
    unsigned v = 0;
    if constexpr (BITS == 3) {
        const int unit = r / 3, wi = r - unit * 3;                               // 32 values in 96 bits, little-endian bit stream (qlinear_cuda.py:144-162)
        for (int f = 0; f < 32; ++f) {
            const int bit = 3 * f - 32 * wi;                                     // position of field f in this word
            if (bit <= -3 || bit >= 32) continue;
            const unsigned x = val(unit * 32 + f);
            v |= bit >= 0 ? (x << bit) : (x >> (-bit));
        }
    } else {
        constexpr int P = 32 / BITS;
#pragma unroll
        for (int f = 0; f < P; ++f) v |= val(r * P + f) << (BITS * f);
    }
    out[(size_t)r * N + n] = v;
}

// one thread = one (strip, group, column): 2 bytes of scale (bit copy) + the zero-point as used (1 byte; 2 bytes at 8 bits, where it reaches 256)
__global__ void __launch_bounds__(256) prepack_decode_consts_kernel(const unsigned* __restrict__ qzeros, const unsigned short* __restrict__ scales, int G, int N,
                                                                    int bits, int zero_mode, unsigned char* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (n >= N) return;
    const int s = n >> 4, col = n & 15, rec_bytes = bits == 8 ? 64 : 48;
    const unsigned maxq = (1u << bits) - 1u;
    const unsigned f = stream_field(qzeros + (size_t)g * (N / 32 * bits), 1, (unsigned)n, bits);
    const unsigned z = zero_mode == GPTQ_ZERO_WRAP ? ((f + 1u) & maxq) : f + 1u;
    unsigned char* rec = out + ((size_t)s * G + g) * rec_bytes;
    *(unsigned short*)(rec + col * 2) = scales[(size_t)g * N + n];
    if (bits == 8) *(unsigned short*)(rec + 32 + col * 2) = (unsigned short)z;
    else rec[32 + col] = (unsigned char)z;
}

template <typename T>
__global__ void __launch_bounds__(256) permute_columns_kernel(const T* __restrict__ x, const int* __restrict__ perm, int M, int K,
                                                              T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int src = perm[i];
    for (int m = blockIdx.y; m < M; m += gridDim.y) out[(size_t)m * K + i] = x[(size_t)m * K + src];
}

// out[m, n] = T( silu(y[m, n]) * y[m, n + N/2] ): the unfused form of the SILU_MUL epilogue (used after the GEMM paths
// and the GEMV kernels that have no fused epilogue); y is the [M, N] = [gate | up] result already rounded to T, which is
// exactly what the reference computes without its fused MLP: act_fn(gate_proj(x)) * up_proj(x).
template <typename T>
__global__ void __launch_bounds__(256) silu_mul_kernel(const T* __restrict__ y, T* __restrict__ out, int M, int N) {
    const int NH = N / 2;
    const size_t total = (size_t)M * NH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / NH, n = i - m * NH;
        const float g = DType<T>::to_f32(y[m * N + n]), u = DType<T>::to_f32(y[m * N + n + NH]);
        out[i] = DType<T>::from_f32(g / (1.f + __expf(-g)) * u);
    }
}

hipError_t launch_silu_mul(const void* y, void* out, int M, int N, int dtype, hipStream_t st) {
    const size_t total = (size_t)M * (N / 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    switch (dtype) {
        case GPTQ_F16: hipLaunchKernelGGL(silu_mul_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const f16*)y, (f16*)out, M, N); break;
        case GPTQ_BF16: hipLaunchKernelGGL(silu_mul_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)y, (bf16*)out, M, N); break;
        default: hipLaunchKernelGGL(silu_mul_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)y, (float*)out, M, N);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
#define GPTQ_BITS_SWITCH(bits, EXPR)                      \
    switch (bits) {                                       \
        case 2: { constexpr int B = 2; EXPR; } break;     \
        case 3: { constexpr int B = 3; EXPR; } break;     \
        case 4: { constexpr int B = 4; EXPR; } break;     \
        case 8: { constexpr int B = 8; EXPR; } break;     \
        default: return hipErrorInvalidValue;             \
    }

hipError_t launch_unpack_weights(const uint32_t* qweight, int K, int N, int bits, uint8_t* w_out, hipStream_t st) {
    const int units = K / unit_vals(bits);
    dim3 grid((N / 4 + 255) / 256, units), block(256);
    GPTQ_BITS_SWITCH(bits, hipLaunchKernelGGL(unpack_weights_kernel<B>, grid, block, 0, st, qweight, units, N, w_out));
    return hipGetLastError();
}

hipError_t launch_unpack_zeros(const uint32_t* qzeros, int G, int N, int bits, int zero_mode, int32_t* z_out, hipStream_t st) {
    dim3 grid((N / 4 + 255) / 256, G), block(256);
    hipLaunchKernelGGL(unpack_zeros_kernel, grid, block, 0, st, qzeros, G, N, bits, zero_mode, z_out);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_dequant_t(const gptq_layer_t& L, void* W_out, hipStream_t st) {
    const int units = L.K / unit_vals(L.bits);
    dim3 grid((L.N / 4 + 255) / 256, units), block(256);
    GPTQ_BITS_SWITCH(L.bits, hipLaunchKernelGGL((dequant_kernel<B, T>), grid, block, 0, st, L.qweight, L.qzeros,
                                                (const T*)L.scales, L.g_idx, units, L.N, L.group_size, L.zero_mode, (T*)W_out));
    return hipGetLastError();
}

hipError_t launch_dequant(const gptq_layer_t& L, void* W_out, hipStream_t st) {
    switch (L.dtype) {
        case GPTQ_F16: return launch_dequant_t<f16>(L, W_out, st);
        case GPTQ_BF16: return launch_dequant_t<bf16>(L, W_out, st);
        case GPTQ_F32: return launch_dequant_t<float>(L, W_out, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pack_weights(const void* W, const void* scale_in, const void* zero_in, const int32_t* g_idx,
                               int K, int N, int bits, int group_size, int w_dtype, int qparam_dtype,
                               uint32_t* qweight_out, void* scales_out, hipStream_t st) {
    dim3 grid((N + 31) / 32, (K + 255) / 256), block(256);       // 32-column x 256-k tiles
    GPTQ_BITS_SWITCH(bits, hipLaunchKernelGGL(pack_weights_kernel<B>, grid, block, 0, st, W, scale_in, zero_in, g_idx, K, N,
                                              group_size, w_dtype, qparam_dtype, qweight_out, scales_out));
    return hipGetLastError();
}

hipError_t launch_pack_zeros(const void* zero_in, int G, int N, int bits, int qparam_dtype, uint32_t* qzeros_out, hipStream_t st) {
    dim3 grid((N / unit_vals(bits) + 255) / 256, G), block(256);
    GPTQ_BITS_SWITCH(bits, hipLaunchKernelGGL(pack_zeros_kernel<B>, grid, block, 0, st, zero_in, G, N, qparam_dtype, qzeros_out));
    return hipGetLastError();
}

hipError_t launch_unprepack_decode(const uint32_t* tiled, int K, int N, int bits, uint32_t* qweight_out, hipStream_t st) {
    const int cke = bits == 8 ? 64 : 128, chunks = (K + cke - 1) / cke, rows = K / 32 * bits;
    const dim3 grid((N + 255) / 256, rows);
    switch (bits) {
        case 4: hipLaunchKernelGGL(unprepack_decode_weights_kernel<4>, grid, dim3(256), 0, st, tiled, K, N, chunks, qweight_out); break;
        case 8: hipLaunchKernelGGL(unprepack_decode_weights_kernel<8>, grid, dim3(256), 0, st, tiled, K, N, chunks, qweight_out); break;
        case 3: hipLaunchKernelGGL(unprepack_decode_weights_kernel<3>, grid, dim3(256), 0, st, tiled, K, N, chunks, qweight_out); break;
        case 2: hipLaunchKernelGGL(unprepack_decode_weights_kernel<2>, grid, dim3(256), 0, st, tiled, K, N, chunks, qweight_out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_prepack_decode(const uint32_t* qweight, const uint32_t* qzeros, const void* scales, int K, int N, int bits, int group_size, int zero_mode,
                                 uint32_t* tiled_out, void* const_out, hipStream_t st) {
    const int cke = bits == 8 ? 64 : 128, chunks = (K + cke - 1) / cke, G = (K + group_size - 1) / group_size;
    const dim3 grid(chunks, N / 16);
    switch (bits) {
        case 4: hipLaunchKernelGGL(prepack_decode_weights_kernel<4>, grid, dim3(256), 0, st, qweight, K, N, chunks, tiled_out); break;
        case 8: hipLaunchKernelGGL(prepack_decode_weights_kernel<8>, grid, dim3(256), 0, st, qweight, K, N, chunks, tiled_out); break;
        case 3: hipLaunchKernelGGL(prepack_decode_weights_kernel<3>, grid, dim3(192), 0, st, qweight, K, N, chunks, tiled_out); break;
        case 2: hipLaunchKernelGGL(prepack_decode_weights_kernel<2>, grid, dim3(128), 0, st, qweight, K, N, chunks, tiled_out); break;
        default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(prepack_decode_consts_kernel, dim3((N + 255) / 256, G), dim3(256), 0, st, qzeros, (const unsigned short*)scales, G, N, bits, zero_mode,
                       (unsigned char*)const_out);
    return hipGetLastError();
}

hipError_t launch_resequence(const uint32_t* qweight, const int32_t* perm, int K, int N, int bits, uint32_t* out, hipStream_t st) {
    const int units = K / unit_vals(bits);
    dim3 grid((N + 255) / 256, units), block(256);
    GPTQ_BITS_SWITCH(bits, hipLaunchKernelGGL(resequence_kernel<B>, grid, block, 0, st, qweight, perm, units, N, out));
    return hipGetLastError();
}

// ---- AWQ ingest (auto_gptq/modeling/_utils.py:525-701) -------------------------------------------------------------
// AutoAWQ packs column 8c + {0,2,4,6,1,3,5,7}[p] into nibble p of word c; so column 8c + i sits at nibble {0,4,1,5,2,6,3,7}[i]
// (the permutation awq_reverse_reorder_int_tensor applies, _utils.py:533-553).  One lane = one AWQ word column (8 output
// columns): 4-byte loads and 16/32-byte stores, both contiguous across the wave.
__device__ __forceinline__ unsigned awq_nibble(unsigned word, int i) {
    const int pos = ((i & 1) << 2) | (i >> 1);               // {0,4,1,5,2,6,3,7}[i]
    return (word >> (4 * pos)) & 15u;
}

// unpack_awq (_utils.py:556-621): w_kn[k, n] = half(w * s) - half(z * s), zeros[g, n] = z; grid.y walks k
__global__ void __launch_bounds__(256) awq_unpack_kernel(const unsigned* __restrict__ aq, const unsigned* __restrict__ az,
                                                         const f16* __restrict__ scales, int K, int N, int group_size,
                                                         f16* __restrict__ w_kn, signed char* __restrict__ zeros) {
    const int NW = N / 8;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= NW) return;
    for (int k = blockIdx.y; k < K; k += gridDim.y) {
        const int g = k / group_size;
        const unsigned qw = aq[(size_t)k * NW + c], qz = az[(size_t)g * NW + c];
        const u32x4 sraw = *(const u32x4*)(scales + (size_t)g * N + c * 8);
        u32x4 o;
        unsigned zb[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned sbits = (sraw[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            const float sf = (float)__builtin_bit_cast(f16, (unsigned short)sbits);
            const unsigned w = awq_nibble(qw, i), z = awq_nibble(qz, i);
            const f16 ws = (f16)((float)w * sf);             // int8 * half -> half (one rounding)
            const f16 zs = (f16)((float)z * sf);
            const f16 d = (f16)((float)ws - (float)zs);      // half - half -> half
            const unsigned db = (unsigned)__builtin_bit_cast(unsigned short, d);
            if (i & 1) o[i >> 1] |= db << 16; else o[i >> 1] = db;
            zb[i >> 2] |= z << (8 * (i & 3));
        }
        *(u32x4*)(w_kn + (size_t)k * N + c * 8) = o;
        if (k % group_size == 0) *(uint2*)(zeros + (size_t)g * N + c * 8) = uint2{zb[0], zb[1]};
    }
}

// unpack_awq + pack_from_tensors as one integer pass: 8 AWQ rows x 1 word column -> 1 GPTQ packed row x 8 columns
__global__ void __launch_bounds__(256) awq_repack_kernel(const unsigned* __restrict__ aq, int K, int N, unsigned* __restrict__ qweight) {
    const int NW = N / 8;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= NW) return;
    for (int r = blockIdx.y; r < K / 8; r += gridDim.y) {
        unsigned a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_nontemporal_load(aq + (size_t)(r * 8 + j) * NW + c);
        unsigned o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned v = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v |= awq_nibble(a[j], i) << (4 * j);
            o[i] = v;
        }
        u32x4* dst = (u32x4*)(qweight + (size_t)r * N + c * 8);
        dst[0] = u32x4{o[0], o[1], o[2], o[3]};
        dst[1] = u32x4{o[4], o[5], o[6], o[7]};
    }
}

// GPTQ qzeros word (g, c): field i = (z[g, 8c + i] - 1) & 15   (pack_from_tensors, _utils.py:679-680)
__global__ void __launch_bounds__(256) awq_repack_zeros_kernel(const unsigned* __restrict__ az, int total, unsigned* __restrict__ qzeros) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const unsigned q = az[t];
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v |= ((awq_nibble(q, i) - 1u) & 15u) << (4 * i);
    qzeros[t] = v;
}

hipError_t launch_awq_unpack(const uint32_t* aq, const uint32_t* az, const void* scales, int K, int N, int group_size, void* w_kn,
                             int8_t* zeros, hipStream_t st) {
    const int NW = N / 8;
    dim3 grid((NW + 255) / 256, K < 4096 ? K : 4096), block(256);
    hipLaunchKernelGGL(awq_unpack_kernel, grid, block, 0, st, aq, az, (const f16*)scales, K, N, group_size, (f16*)w_kn, (signed char*)zeros);
    return hipGetLastError();
}

hipError_t launch_awq_repack(const uint32_t* aq, const uint32_t* az, int K, int N, int group_size, uint32_t* qweight, uint32_t* qzeros,
                             hipStream_t st) {
    const int NW = N / 8, R = K / 8;
    dim3 grid((NW + 255) / 256, R < 4096 ? R : 4096), block(256);
    hipLaunchKernelGGL(awq_repack_kernel, grid, block, 0, st, aq, K, N, qweight);
    const int total = (K / group_size) * NW;
    hipLaunchKernelGGL(awq_repack_zeros_kernel, dim3((total + 255) / 256), dim3(256), 0, st, az, total, qzeros);
    return hipGetLastError();
}

hipError_t launch_permute_columns(const void* x, const int32_t* perm, int M, int K, int dtype, void* x_out, hipStream_t st) {
    // the LDS-staged row kernel runs one workgroup per row of x: for a few long rows (decode of act-order layers, M <= 4) the flat
    // one-thread-per-element gather below has K/256 x M workgroups instead (K = 28672, M = 1: ~7 us against ~2.5)
    if (dtype != GPTQ_F32 && K % 8 == 0 && (size_t)K * 2 <= 64 * 1024 && (M >= 64 || K <= 4096)) return launch_permute_rows16(x, perm, M, K, x_out, st);
    dim3 grid((K + 255) / 256, M < 1024 ? M : 1024), block(256);
    if (dtype == GPTQ_F32)
        hipLaunchKernelGGL(permute_columns_kernel<float>, grid, block, 0, st, (const float*)x, perm, M, K, (float*)x_out);
    else
        hipLaunchKernelGGL(permute_columns_kernel<unsigned short>, grid, block, 0, st, (const unsigned short*)x, perm, M, K,
                           (unsigned short*)x_out);
    return hipGetLastError();
}

}  // namespace gptq
