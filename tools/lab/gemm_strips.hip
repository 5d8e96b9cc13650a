// gemm_strips.hip -- batched decode, round 5: 5 .. 64 rows of x on the DECODE COPY (gptq_prepack_decode), 4-bit fp16 / bf16.
//
// Role in the reference: the "custom kernel" band between the one-row kernels and the dequantise-then-GEMM fallbacks -- exllamav2 serves up to
// MAX_Q_GEMM_ROWS = 50 rows from the same re-laid matrix as one row (autogptq_extension/exllamav2/cuda/q_gemm.cu:118, config.h:4,
// q_gemm_kernel_gptq.cuh:39-194), cuda / cuda_old use the fused kernel below kernel_switch_threshold = 128 rows (qlinear_cuda.py:34,212).
//
// Why: rounds 2-4 served this band from the checkpoint rows (gemm_strip16 / gemm_stream64 / gemm_mid: 64-byte or 256-byte row segments 16 KiB apart) and
// every 16-column strip fetched its own copy of x from the L2 -- a CU pulls a shared operand out of the L2 at ~50 GB/s (profiles/r03_xfetch_lab.log), so
// M x K x 2 bytes per strip was what the time grew with (M = 16 on 4096^2: 7.85 us against 4.54 at one row).  Here
//   * a workgroup owns FOUR adjacent strips (64 columns) x one K slice and stages its part of x ONCE (LDS DMA, issued first);
//   * the weights are the decode copy: a strip's chunk is one contiguous KiB = one wave load, a lane (k-slot s, column c) holds 4 words = 32 consecutive k
//     of ONE column in pair order -- word j of every lane is the B operand of ONE v_mfma_f32_16x16x32 (lane = column l & 15, k-group l >> 4 = k-slot:
//     k = 32 s + 8 j + 0..7), whose A operand is the 16 bytes x[row l & 15][chunk k0 + 32 s + 8 j ..] of the RAW staged x (one conflict-free ds_read_b128);
//   * 16 waves = 4 strips x 4 quarters of the slice's chunks: no barrier in the K loop; the quarters meet through LDS (the dead x tile), K slices of a
//     column group through {fp32, tag} granules as in the decode kernel (gemv_shared.cuh: stream_finish), one epoch word per column group;
//   * dequant: the exact magic-number form, scale applied in the fragment (the reference's W = scales * (w - z), one rounding): group sizes from 32.
// Act-order layers: the copy holds the re-sequenced rows; x is permuted (natural order) by the pre-pass, as for the prefill kernels.
#include <type_traits>

#include "common.cuh"
#include "gemv_shared.cuh"
#include "launch.h"

namespace gptq {
namespace strips {

struct StripsParams {
    const unsigned* tq;          // qweight_tiled
    const void* cst;             // qconst_tiled: [strip][group][48 B] = 16 scales + 16 one-byte zero-points as used
    const void* bias;
    void* out;
    const void* x;
    unsigned long long* gran;    // [ksplit - 1][M][N] {fp32, tag} granules of K slices 1 ..
    unsigned* epochs;            // workspace header, epoch half: one word per column group (monotonic, bumped by the owner slice)
    unsigned* err;
    int M, K, N, nstrips, chunks, cps, ksplit, G, gshift, xstride, cpad;
    unsigned max_spins;
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

// one column: (scale bits, zero-point as used) -> the B fragment of one stored word = 8 consecutive k (pair order of the copy): exact w - z in packed fp16
// (magic number 0x6400), times the scale -- fp16: one packed multiply = the reference's fp16 W; bf16: exact fp32 product, one rounding to bf16
template <typename T> struct DeqCol;
template <> struct DeqCol<f16> {
    f16x2 s2, c1, c2;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);                              // -(1024 + z)
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;                                                             // -(64 + z)
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        u32x4 o;
        o[0] = __builtin_bit_cast(unsigned, (as_f16x2((q & 0x000f000fu) | 0x64006400u) + c1) * s2);
        o[1] = __builtin_bit_cast(unsigned, (as_f16x2((q & 0x00f000f0u) | 0x64006400u) * r16 + c2) * s2);
        o[2] = __builtin_bit_cast(unsigned, (as_f16x2((q8 & 0x000f000fu) | 0x64006400u) + c1) * s2);
        o[3] = __builtin_bit_cast(unsigned, (as_f16x2((q8 & 0x00f000f0u) | 0x64006400u) * r16 + c2) * s2);
        return o;
    }
};
template <> struct DeqCol<bf16> {
    f16x2 c1, c2;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sbits);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;
    }
    static __device__ __forceinline__ unsigned scaled_pair(f16x2 h, float sc) {
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        float lo, hi;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(sc));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(hb), "v"(sc));
        const bf16x2 v = {(bf16)lo, (bf16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        u32x4 o;
        o[0] = scaled_pair(as_f16x2((q & 0x000f000fu) | 0x64006400u) + c1, s);
        o[1] = scaled_pair(as_f16x2((q & 0x00f000f0u) | 0x64006400u) * r16 + c2, s);
        o[2] = scaled_pair(as_f16x2((q8 & 0x000f000fu) | 0x64006400u) + c1, s);
        o[3] = scaled_pair(as_f16x2((q8 & 0x00f000f0u) | 0x64006400u) * r16 + c2, s);
        return o;
    }
};

constexpr int SPW = 4;                       // strips per workgroup (64 columns)
constexpr int KQ = 4;                        // K quarters of the slice (waves per strip)
constexpr int U = 4;                         // chunks in flight per wave

// RT: 16-row tiles of x (M <= 16 RT)
template <typename T, int RT>
__global__ void __launch_bounds__(1024) gemm_strips_kernel(StripsParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kslot = lane >> 4;
    const int st = wave & (SPW - 1), kq = wave >> 2;
    int cgi = blockIdx.x, ks = 0;
    if (p.ksplit != 1) { cgi = (int)blockIdx.x / p.ksplit; ks = (int)blockIdx.x - cgi * p.ksplit; }
    const int strip = cgi * SPW + st;
    const bool strip_ok = strip < p.nstrips;
    const int strip_l = strip_ok ? strip : p.nstrips - 1;                          // a missing strip of the last column group: valid loads, nothing emitted
    const int cb = ks * p.cps, ce = min(cb + p.cps, p.chunks);                     // this slice's chunks
    const int kbeg = cb * 128, klen = (ce - cb) * 128;
    // LDS: [x: 16 RT rows of klen values, row stride klen * 2 + 16 bytes (the 16 rows of an A fragment read hit 16 different bank groups)]
    //      [constants: 4 strips x G x 48 bytes]; the x tile is reused for the cross-wave sums behind the K loop
    char* const xs = smem;
    char* const cs = smem + (size_t)(16 * RT) * p.xstride;
    const unsigned xs_lds = lds_addr_of(xs), cs_lds = lds_addr_of(cs);
    // ---- stage x (all waves) and the constants (one wave per strip) by LDS DMA: issued first, waited for behind the first weight burst
    {
        const int pieces = klen >> 3;                                              // 16-byte pieces per row (a multiple of 16)
        const int nblk = (pieces + 63) >> 6;
        const int units = 16 * RT * nblk;
        for (int u = wave; u < units; u += 16) {
            const int m = u / nblk, pc0 = (u - m * nblk) << 6;
            const char* xr = (const char*)p.x + ((size_t)min(m, p.M - 1) * p.K + kbeg) * 2;
            if (pc0 + lane < pieces) lds_dma16(xr + (size_t)(pc0 + lane) * 16, xs_lds + (unsigned)m * (unsigned)p.xstride + (unsigned)pc0 * 16u);
        }
        if (kq == 0) {
            const char* cg = (const char*)p.cst + (size_t)strip_l * p.G * 48;
            const int cpieces = (p.G * 48) >> 4;
            for (int pc0 = 0; pc0 < cpieces; pc0 += 64)
                if (pc0 + lane < cpieces) dma16_nt(cg + (size_t)(pc0 + lane) * 16, __builtin_amdgcn_readfirstlane(cs_lds + (unsigned)st * (unsigned)p.cpad + (unsigned)pc0 * 16u));
        }
    }
    // ---- this wave's chunks: quarter kq of the slice
    const int per = (ce - cb + KQ - 1) / KQ;
    const int c_lo = cb + kq * per, c_hi = min(c_lo + per, ce);
    const char* const tb = (const char*)p.tq + (size_t)strip_l * p.chunks * 1024 + (size_t)lane * 16;
    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* const xa = xs + (size_t)col * p.xstride + kslot * 64;               // A operand: lane (row col, k-group kslot)
    const char* const cl = cs + (size_t)st * p.cpad;
    u32x4 q[U];
    const int c_last = max(c_hi - 1, cb);                                           // (an empty quarter loads a valid chunk and computes nothing)
#pragma unroll
    for (int j = 0; j < U; ++j) q[j] = __builtin_nontemporal_load((const u32x4*)(tb + (size_t)min(c_lo + j, c_last) * 1024));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");                        // the staging DMAs are OLDER than the U loads just issued
    __syncthreads();
    for (int c0 = c_lo; c0 < c_hi; c0 += U) {
        if (c0 != c_lo) {
#pragma unroll
            for (int j = 0; j < U; ++j) q[j] = __builtin_nontemporal_load((const u32x4*)(tb + (size_t)min(c0 + j, c_last) * 1024));
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (c0 + j >= c_hi) break;                                              // wave-uniform
            const int cc = c0 + j;
            const int g = min(((cc * 128 + kslot * 32) >> 5) >> p.gshift, p.G - 1);
            const char* cp = cl + g * 48;
            const unsigned sraw = *(const unsigned short*)(cp + col * 2);
            const unsigned z = *(const unsigned char*)(cp + 32 + col);
            DeqCol<T> dq;
            dq.setup(sraw, z);
            u32x4 b[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) b[w] = dq.frag(q[j][w]);
            const char* xc = xa + (size_t)(cc - cb) * 256;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const char* xr = xc + (size_t)(rt * 16) * p.xstride;
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[rt] = Mma16<T>::run(*(const u32x4*)(xr + w * 16), b[w], acc[rt]);
            }
        }
    }
    // ---- K quarters through LDS (the x tile is dead behind the barrier), then write, or publish / combine the K slices through granules
    __syncthreads();
    f32x4* const red = (f32x4*)smem;                                               // [wave][rt][lane]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) red[(wave * RT + rt) * 64 + lane] = acc[rt];
    __syncthreads();
    unsigned tag = 0;
    if (p.ksplit > 1) {
        unsigned ep;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ep) : "s"(p.epochs + cgi) : "memory");
        tag = 0x7FE00000u | ((ep + 1u) & 0x1FFFFFu);
    }
    const size_t slab = (size_t)p.M * p.N;
    bool gave_up = false;
    for (int idx = tid; idx < SPW * RT * 64; idx += 1024) {
        const int s4 = idx / (RT * 64), rem = idx - s4 * (RT * 64), rt = rem >> 6, ln = rem & 63;
        f32x4 t = red[((0 * SPW + s4) * RT + rt) * 64 + ln];
#pragma unroll
        for (int k = 1; k < KQ; ++k) t += red[((k * SPW + s4) * RT + rt) * 64 + ln];      // fixed order
        const int n = (cgi * SPW + s4) * 16 + (ln & 15);
        if (cgi * SPW + s4 >= p.nstrips) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                              // C/D layout: row = 4 * (lane >> 4) + r, column = lane & 15
            const int m = rt * 16 + 4 * (ln >> 4) + r;
            if (m >= p.M) continue;
            float v = t[r];
            if (p.ksplit > 1) {
                const size_t at = (size_t)m * p.N + n;
                if (ks != 0) {
                    const unsigned long long g8 = (unsigned long long)as_u32(v) | ((unsigned long long)tag << 32);
                    __hip_atomic_store(p.gran + (size_t)(ks - 1) * slab + at, g8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
                }
                constexpr int KMAX = 8;
                unsigned long long gv[KMAX - 1];
                unsigned pending = (1u << (p.ksplit - 1)) - 1u;
                for (unsigned spins = 0; pending; ++spins) {
#pragma unroll
                    for (int k = 0; k < KMAX - 1; ++k)
                        if (pending & (1u << k)) gv[k] = __hip_atomic_load(p.gran + (size_t)k * slab + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < KMAX - 1; ++k)
                        if ((pending & (1u << k)) && (unsigned)(gv[k] >> 32) == tag) pending &= ~(1u << k);
                    if (pending && spins > p.max_spins) { gave_up = true; break; }
                    if (pending) __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int k = 0; k < KMAX - 1; ++k)
                    if (k < p.ksplit - 1) v += as_f32((unsigned)(gv[k] & 0xffffffffu));      // slice order: bit-reproducible
#pragma unroll
                for (int k = 0; k < KMAX - 1; ++k)                                  // consumed granules are cleared: no valid tag survives a launch
                    if (k < p.ksplit - 1) __hip_atomic_store(p.gran + (size_t)k * slab + at, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (p.bias) v += DType<T>::to_f32(((const T*)p.bias)[n]);
            ((T*)p.out)[(size_t)m * p.N + n] = DType<T>::from_f32(v);
        }
    }
    if (p.ksplit > 1 && ks == 0) {
        if (gave_up) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                                            // every wave of the owner has its granules: every producer wave has read the epoch
        if (tid == 0) __hip_atomic_store(p.epochs + cgi, (tag & 0x1FFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace strips

// ---- host side -------------------------------------------------------------------------------------------------------------------------------
bool strips_layer_ok(const gptq_layer_t& L) {
    if (L.bits != 4 || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16) || L.epilogue != GPTQ_EPI_NONE) return false;
    if (!L.qweight_tiled || !L.qconst_tiled || L.tiled_cols != GPTQ_STRIP_COLS) return false;
    if (L.g_idx != nullptr && !(L.perm && L.qweight_seq)) return false;
    if (L.K % 128 || L.N % GPTQ_STRIP_COLS) return false;
    const int gu = L.group_size / 32;
    return L.group_size % 32 == 0 && (L.group_size >= L.K || (gu & (gu - 1)) == 0);
}

// Provisional (before the sweep of tools/strips_ab.py): never by default.
bool strips_pays(const gptq_layer_t& L, int M) { (void)L; (void)M; return false; }

StripsPlan plan_strips(const gptq_layer_t& L, int M, const gptq_tuning_t* tune) {
    StripsPlan pl{};
    if (!strips_layer_ok(L) || M < 1 || M > 64) return pl;
    pl.rt = M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 48 ? 3 : 4));
    pl.nstrips = L.N / GPTQ_STRIP_COLS;
    pl.cgroups = (pl.nstrips + strips::SPW - 1) / strips::SPW;
    pl.chunks = L.K / 128;
    pl.groups = (L.K + L.group_size - 1) / L.group_size;
    pl.cpad = (pl.groups * 48 + 15) & ~15;
    // K slices: enough workgroups to fill the chip (one 16-wave workgroup per CU) and a slice of x that fits the LDS next to the constants
    const size_t lds_cap = 144 * 1024;
    auto lds_need = [&](int cps) {
        const size_t xt = (size_t)(16 * pl.rt) * ((size_t)cps * 256 + 16), red = (size_t)16 * pl.rt * 1024;
        return (xt > red ? xt : red) + (size_t)strips::SPW * pl.cpad + 16;
    };
    int ks = (tune && tune->ksplit > 0) ? tune->ksplit : 0;
    if (!ks) {
        ks = 1;
        while (ks < 8 && pl.cgroups * ks < 224 && pl.chunks / (ks * 2) >= 4) ks *= 2;
    }
    while (ks < 8 && lds_need((pl.chunks + ks - 1) / ks) > lds_cap) ks *= 2;
    if (ks > 8 || ks > pl.chunks) return pl;
    pl.cps = (pl.chunks + ks - 1) / ks;
    pl.ksplit = (pl.chunks + pl.cps - 1) / pl.cps;
    if (lds_need(pl.cps) > lds_cap) return pl;
    if ((size_t)pl.cgroups * 4 > WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES - WS_HEADER_EPOCH_OFFSET) return pl;
    pl.xstride = pl.cps * 256 + 16;
    pl.lds_bytes = lds_need(pl.cps);
    pl.use_seq = L.g_idx != nullptr;
    pl.xperm_bytes = pl.use_seq ? (((size_t)M * L.K * 2 + 255) / 256 * 256) : 0;
    pl.partial_bytes = pl.ksplit > 1 ? (size_t)(pl.ksplit - 1) * M * L.N * 8 : 0;
    pl.ok = true;
    return pl;
}

template <typename T, int RT>
static hipError_t grant_strips() {
    return hipFuncSetAttribute((const void*)strips::gemm_strips_kernel<T, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t init_gemm_strips_device() {
    hipError_t e = grant_strips<f16, 1>();
    if (e == hipSuccess) e = grant_strips<f16, 2>();
    if (e == hipSuccess) e = grant_strips<f16, 3>();
    if (e == hipSuccess) e = grant_strips<f16, 4>();
    if (e == hipSuccess) e = grant_strips<bf16, 1>();
    if (e == hipSuccess) e = grant_strips<bf16, 2>();
    if (e == hipSuccess) e = grant_strips<bf16, 3>();
    if (e == hipSuccess) e = grant_strips<bf16, 4>();
    return e;
}

template <typename T>
static hipError_t launch_strips_t(const StripsPlan& pl, const strips::StripsParams& p, hipStream_t st) {
    const dim3 grid(pl.cgroups * pl.ksplit), block(1024);
    switch (pl.rt) {
        case 1: hipLaunchKernelGGL((strips::gemm_strips_kernel<T, 1>), grid, block, pl.lds_bytes, st, p); break;
        case 2: hipLaunchKernelGGL((strips::gemm_strips_kernel<T, 2>), grid, block, pl.lds_bytes, st, p); break;
        case 3: hipLaunchKernelGGL((strips::gemm_strips_kernel<T, 3>), grid, block, pl.lds_bytes, st, p); break;
        case 4: hipLaunchKernelGGL((strips::gemm_strips_kernel<T, 4>), grid, block, pl.lds_bytes, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// x: the activations the kernel reads (act-order layers: already permuted in natural order); gran: (ksplit - 1) * M * N granules when ksplit > 1
hipError_t launch_strips(const gptq_layer_t& L, const StripsPlan& pl, const void* x, void* out, int M, void* ws_header, void* gran, hipStream_t st) {
    if (!pl.ok) return hipErrorInvalidValue;
    if (pl.ksplit > 1 && (!ws_header || !gran)) return hipErrorInvalidValue;
    strips::StripsParams p{};
    p.tq = L.qweight_tiled; p.cst = L.qconst_tiled; p.bias = L.bias; p.out = out; p.x = x;
    p.gran = (unsigned long long*)gran;
    p.epochs = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_EPOCH_OFFSET) : nullptr;
    p.err = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2 : nullptr;
    p.M = M; p.K = L.K; p.N = L.N; p.nstrips = pl.nstrips; p.chunks = pl.chunks; p.cps = pl.cps; p.ksplit = pl.ksplit; p.G = pl.groups;
    p.gshift = L.group_size >= L.K ? 26 : __builtin_ctz((unsigned)(L.group_size / 32));
    p.xstride = pl.xstride; p.cpad = pl.cpad;
    p.max_spins = 1u << 20;
    return L.dtype == GPTQ_BF16 ? launch_strips_t<bf16>(pl, p, st) : launch_strips_t<f16>(pl, p, st);
}

}  // namespace gptq
