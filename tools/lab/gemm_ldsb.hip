// gemm_ldsb.hip -- prefill GEMM with the dequantised weights SHARED THROUGH LDS (4-bit, fp16 / bf16, large M).
//
// Role in the reference: Marlin's main loop (autogptq_extension/marlin/marlin_cuda_kernel.cu:414-470 fetch_to_shared / fetch_to_registers,
// :733 STAGES) and the dequant + cublasHgemm fallbacks (exllama q4_matmul.cu:225-260, exllamav2 q_gemm.cu:104-181).  Nothing is derived from them.
//
// Why a second tiled kernel: gemm_kernel (gemm.hip) feeds each lane's own weight words straight to the matrix core -- exact and traffic-free,
// but every dequantised word is reused by only the 4 row tiles of ONE wave: 4.7 VALU per MFMA, matrix pipe 45-50 % busy (profiles/r02_gemm_pmc.txt).
// Here a 256 x 128 workgroup tile dequantises each weight ONCE per workgroup into LDS (natural k order, one ds_write_b128 per word) and all four
// waves (2 x 2, 128 x 64 each) read it back as B fragments: each word is reused by 8 row tiles -> ~1.9 dequant VALU per MFMA, and because B is in
// natural order x is staged VERBATIM by LDS DMA (global_load_lds_dwordx4, XOR-swizzled slots) for every layer, not only the act-order ones.
//   LDS per workgroup: A 2 x 256 x 128 B = 64 KiB, B 2 x 8 x 128 x 16 B = 32 KiB.
//   per K-step (64 deep) and wave: 32 MFMA 32x32x16, 16 + 8 ds_read_b128, 4 ds_write_b128, 8 DMA instructions, 4 + 2 loads.
#include <type_traits>

#include "common.cuh"
#include "launch.h"
#include "lab_launch.h"

namespace gptq {

namespace {

struct LdsbParams {
    const unsigned* qweight;
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
    const void* x;
    void* out;
    int M, K, N, zero_mode;
    int nbm, nbn, ksteps;
    int lg_spg32;      // log2(group_size / 32) (a power of two)
};

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma32;
template <> struct Mma32<f16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    }
};
template <> struct Mma32<bf16> {
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    }
};

// 16 bytes per lane global -> LDS, source = scalar base + 32-bit per-lane offset (no vector address arithmetic per K-step)
__device__ __forceinline__ void dma16_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// one column's group constants -> exact packed dequant of one 4-bit word (8 consecutive k) in NATURAL k order
template <typename T> struct Deq1N;
template <> struct Deq1N<f16> {
    f16x2 s2, c1, c2;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);             // -(1024 + z)
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;                                           // -(64 + z)
    }
    __device__ __forceinline__ u32x4 word(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = (as_f16x2((q & 0x000f000fu) | 0x64006400u) + c1) * s2;          // k0,k4 : s * (w - z), one rounding
        const f16x2 h1 = (as_f16x2((q & 0x00f000f0u) | 0x64006400u) * r16 + c2) * s2;    // k1,k5
        const f16x2 h2 = (as_f16x2((q8 & 0x000f000fu) | 0x64006400u) + c1) * s2;         // k2,k6
        const f16x2 h3 = (as_f16x2((q8 & 0x00f000f0u) | 0x64006400u) * r16 + c2) * s2;   // k3,k7
        const unsigned a = __builtin_bit_cast(unsigned, h0), b = __builtin_bit_cast(unsigned, h1), c = __builtin_bit_cast(unsigned, h2),
                       d = __builtin_bit_cast(unsigned, h3);
        // (k0,k4)(k1,k5)(k2,k6)(k3,k7) -> (k0,k1)(k2,k3)(k4,k5)(k6,k7)
        return u32x4{__builtin_amdgcn_perm(b, a, 0x05040100u), __builtin_amdgcn_perm(d, c, 0x05040100u), __builtin_amdgcn_perm(b, a, 0x07060302u),
                     __builtin_amdgcn_perm(d, c, 0x07060302u)};
    }
};
template <> struct Deq1N<bf16> {
    f16x2 c1, c2;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s = as_f32(sbits << 16);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;
    }
    // bf16(s * (w - z)): w - z exact in fp16, the product exact in fp32, one rounding to bf16 -- the reference's scales * (weight - zeros)
    static __device__ __forceinline__ void scaled(f16x2 h, float sc, float& lo, float& hi) {
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(sc));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(hb), "v"(sc));
    }
    __device__ __forceinline__ u32x4 word(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        const f16x2 h0 = as_f16x2((q & 0x000f000fu) | 0x64006400u) + c1;
        const f16x2 h1 = as_f16x2((q & 0x00f000f0u) | 0x64006400u) * r16 + c2;
        const f16x2 h2 = as_f16x2((q8 & 0x000f000fu) | 0x64006400u) + c1;
        const f16x2 h3 = as_f16x2((q8 & 0x00f000f0u) | 0x64006400u) * r16 + c2;
        float k0, k4, k1, k5, k2, k6, k3, k7;
        scaled(h0, s, k0, k4); scaled(h1, s, k1, k5); scaled(h2, s, k2, k6); scaled(h3, s, k3, k7);
        const bf16x2 p0 = {(bf16)k0, (bf16)k1}, p1 = {(bf16)k2, (bf16)k3}, p2 = {(bf16)k4, (bf16)k5}, p3 = {(bf16)k6, (bf16)k7};
        return u32x4{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
    }
};

constexpr int LB_BM = 256, LB_BN = 128;
template <int BK> constexpr int lb_a_bytes() { return LB_BM * BK * 2; }           // x tile per stage
template <int BK> constexpr int lb_b_bytes() { return (BK / 8) * LB_BN * 16; }    // dequantised weight tile per stage
template <int BK, int KG> constexpr int lb_lds_bytes() {
    const int stages = KG * 2 * (lb_a_bytes<BK>() + lb_b_bytes<BK>());
    return (KG == 2 && stages < 65536) ? 65536 : stages;                           // KG = 2: the 64 KiB exchange area of the final sum aliases the stages
}

// Workgroup = KG groups of 4 waves; a group is 2 x 2 waves over the 256 x 128 tile: wave (wm, wn) owns rows wm * 128 .. + 127 (4 row tiles) and
// columns wn * 64 .. + 63 (2 column tiles: tile nt = the columns of parity nt, so a lane's two accumulators are ADJACENT columns and the
// epilogue stores 4 bytes).  KG = 2 (launches with at most one tile per CU): the second group works on the other half of K with its own stages
// (two waves per SIMD: one group's DMA issue, dequant and barrier waits run under the other's MFMAs), the halves are summed through LDS at the end.
// BK = 32 keeps a group's stages at 48 KiB: two workgroups (KG = 1) or two groups (KG = 2) per CU.
// x tile rows are BK * 2 bytes, DMA-written linearly; 16-byte slot s of row r holds k-chunk s ^ swz(r): swz(r) = (r >> 1) & 7 for 128-byte rows,
// (r >> 2) & 3 for 64-byte rows -- the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte bank groups.
// ABL (tools/ldsblab only; results are wrong by construction): bit 0 = no dequant math (raw words stored), bit 1 = no x DMA after the prologue,
// bit 2 = no per-step barrier, bit 3 = no weight-word loads after the prologue, bit 4 = no B store at all
template <typename T, int BK, int KG, int ABL = 0>
__global__ void __launch_bounds__(256 * KG, 2 / KG) gemm_ldsb_kernel(LdsbParams p) {
    constexpr int KS = BK / 16;                        // MFMA k-steps per K-step
    constexpr int KB = BK / 8;                         // k-blocks (packed rows) per K-step
    constexpr int WPT = KB * LB_BN / 256;              // weight words per thread and K-step (4 or 2)
    constexpr int A_BYTES = lb_a_bytes<BK>(), B_BYTES = lb_b_bytes<BK>();
    constexpr int ROWB = BK * 2;                       // bytes per x row in LDS
    constexpr int CPR = BK / 8;                        // 16-byte chunks per row (8 or 4)
    constexpr int NDMA = A_BYTES / 1024 / 4;           // DMA instructions per wave and K-step (8 or 4)
    constexpr int NLD = WPT + 2;                       // visible loads per K-step (weight words + scale + zero word)
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    char* const smem = smem_all + (size_t)kg * (2 * (A_BYTES + B_BYTES));
    char* const sA = smem;                             // [2][256 rows][CPR slots x 16 B]
    char* const sB = smem + 2 * A_BYTES;               // [2][KB k-blocks][128 positions][16 B]
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // tile order: as gemm_kernel -- an XCD's contiguous run of logical ids is a compact patch 8 column tiles wide
    const int L = xcd_remap(blockIdx.x, p.nbm * p.nbn);
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
    const int m0 = bm * LB_BM, n0 = bn * LB_BN;
    int kt0 = 0, kt1 = p.ksteps;
    if constexpr (KG == 2) {                           // the planner only picks KG = 2 for an even step count
        const int hs = p.ksteps >> 1;
        kt0 = kg * hs;
        kt1 = kt0 + hs;
    }

    // ---- A staging by DMA: instruction i of wave w fills LDS chunks [(i * 4 + w) * 64, + 64): chunk c = row c / CPR, slot c % CPR
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    unsigned a_off[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int c = (i * 4 + wave) * 64 + lane;
        const int row = c / CPR, slot = c % CPR;
        const int src = slot ^ swz(row);
        const int mrow = min(m0 + row, p.M - 1) - m0;                       // rows past M re-read the last row (their results are not stored)
        a_off[i] = (unsigned)mrow * (unsigned)p.K * 2u + (unsigned)src * 16u;
    }
    const char* const a_base = (const char*)p.x + (size_t)m0 * p.K * 2;
    const unsigned sA_lds = lds_addr_of(sA);
    auto dma_a = [&](int kt, int buf) __attribute__((always_inline)) {
        const char* base = a_base + (size_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma16_sv(base, a_off[i], sA_lds + (unsigned)buf * A_BYTES + (unsigned)(i * 4 + wave) * 1024u);
    };

    // ---- B production: thread t dequantises the words (k-block kq * WPT + j, column t & 127), j < WPT, of every K-step
    const int bcol = tid & 127, kq = tid >> 7;
    const int ncol = min(n0 + bcol, p.N - 1);
    const int G = p.K >> (5 + p.lg_spg32);
    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qweight, 0, (int)((size_t)(p.K / 8) * p.N * 4), 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, 0, (int)((size_t)G * p.N * 2), 0x00020000);
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.qzeros, 0, (int)((size_t)G * (p.N / 8) * 4), 0x00020000);
    const unsigned q_lane_off = ((unsigned)(kq * WPT) * (unsigned)p.N + (unsigned)ncol) * 4u;
    const unsigned s_lane_off = ((unsigned)ncol >> 1) * 4u, s_lane_sh = ((unsigned)ncol & 1u) * 16u;
    const unsigned z_lane_off = ((unsigned)ncol >> 3) * 4u, z_lane_sh = ((unsigned)ncol & 7u) * 4u;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;
    const int lg_spg = p.lg_spg32 - (BK == 64 ? 1 : 0);                    // log2(K-steps per group)
    struct WRaw { unsigned q[WPT]; unsigned s, z; };
    auto load_w = [&](int kt, WRaw& w) __attribute__((always_inline)) {
        const int ktc = min(kt, kt1 - 1);
#pragma unroll
        for (int j = 0; j < WPT; ++j) w.q[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_q, q_lane_off, (unsigned)((size_t)(ktc * KB + j) * p.N * 4), 0);
        const unsigned g = (unsigned)ktc >> lg_spg;
        w.s = __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, s_lane_off, g * (unsigned)p.N * 2u, 0);
        w.z = __builtin_amdgcn_raw_buffer_load_b32(rsrc_z, z_lane_off, g * (unsigned)(p.N / 8) * 4u, 0);
    };
    // LDS position of this thread's column inside a k-block row: [column half][parity][index]
    const unsigned b_wpos = (unsigned)((bcol >> 6) * 64 + (bcol & 1) * 32 + ((bcol & 63) >> 1));
    auto store_b = [&](int buf, const WRaw& w, int j) __attribute__((always_inline)) {
        if constexpr (ABL & 16) return;
        char* dst = sB + buf * B_BYTES + ((kq * WPT + j) * LB_BN + b_wpos) * 16;
        if constexpr (ABL & 1) {
            *(u32x4*)dst = u32x4{w.q[j], w.q[j] ^ 0x11111111u, w.s, w.z};
            return;
        }
        Deq1N<T> dq;
        dq.setup((w.s >> s_lane_sh) & 0xffffu, (((w.z >> z_lane_sh) & 15u) + 1u) & zmask);
        *(u32x4*)dst = dq.word(w.q[j]);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int a_swz = swz(l31);                                             // the tiles' first rows are multiples of 32: swz(row) = swz(row & 31)
    const char* const a_rd = sA + (wm * 128 + l31) * ROWB;                 // + buf * A_BYTES + mt * 32 * ROWB + ((2 ks + half) ^ a_swz) * 16
    const char* const b_rd = sB + (wn * 64 + l31) * 16 + half * (LB_BN * 16);   // + buf * B_BYTES + (2 ks) * LB_BN * 16 + nt * 32 * 16

    // ---- prologue: stage 0 of A and B, the words of step 1 in flight
    WRaw w0, w1;
    dma_a(kt0, 0);
    load_w(kt0, w0);
    load_w(kt0 + 1, w1);
#pragma unroll
    for (int j = 0; j < WPT; ++j) store_b(0, w0, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");             // the DMAs (older than the loads of w1) have landed
    __syncthreads();

    // one K-step: MFMAs on stage BUF, meanwhile stage BUF ^ 1 is produced (x tile by DMA, B tile from the words in w_use)
    auto step = [&](int kt, auto bufc, WRaw& w_use, WRaw& w_fill) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        // claim the words of the NEXT stage (requested a whole step ago) before anything new is issued: the compiler's wait for them lands
        // here and is exact; every wait it would emit later, with the DMAs below in flight, would be too strict by their number
#pragma unroll
        for (int j = 0; j < WPT; ++j) asm volatile("" ::"v"(w_use.q[j]));
        asm volatile("" ::"v"(w_use.s), "v"(w_use.z));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 2)) dma_a(min(kt + 1, kt1 - 1), BUF ^ 1);
        WRaw wn_ = w_use;
        if constexpr (!(ABL & 8)) load_w(kt + 2, wn_);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 a[2][4], b[2][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[0][mt] = *(const u32x4*)(a_rd + BUF * A_BYTES + mt * 32 * ROWB + ((half ^ a_swz) * 16));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[0][nt] = *(const u32x4*)(b_rd + BUF * B_BYTES + nt * 512);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    a[(ks + 1) & 1][mt] = *(const u32x4*)(a_rd + BUF * A_BYTES + mt * 32 * ROWB + ((((ks + 1) * 2 + half) ^ a_swz) * 16));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) b[(ks + 1) & 1][nt] = *(const u32x4*)(b_rd + BUF * B_BYTES + (ks + 1) * 2 * (LB_BN * 16) + nt * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = ks * WPT / KS; j < (ks + 1) * WPT / KS; ++j) store_b(BUF ^ 1, w_use, j);     // the next B tile, spread over the MFMA groups
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Mma32<T>::run(a[ks & 1][mt], b[ks & 1][nt], acc[mt][nt]);
        }
        w_fill = wn_;
        if constexpr (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");   // this step's DMAs have landed (the loads behind them may fly on)
        if constexpr (!(ABL & 4)) __syncthreads();
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
        step(kt, std::integral_constant<int, 0>{}, w1, w0);                // during step kt the words of kt + 1 (w1) become B[1]; w0 <- kt + 2
        if (kt + 1 < kt1) step(kt + 1, std::integral_constant<int, 1>{}, w0, w1);
    }

    if constexpr (KG == 2) {
        // sum the two K halves through LDS (the stages are dead after the last barrier): group 1 hands row tiles 0-1 to group 0, then group 0
        // hands row tiles 2-3 to group 1; each group stores the half it completed
        float4* ex = (float4*)smem_all;                // [(mt, nt, quad)][256 threads] float4: lane-contiguous, conflict free; 64 KiB
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (kg != pass) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x16& v = acc[pass * 2 + mt][nt];
                            ex[((mt * 2 + nt) * 4 + q) * 256 + tid] = float4{v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
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
                            f32x16& o = acc[pass * 2 + mt][nt];
                            o[q * 4] += v.x; o[q * 4 + 1] += v.y; o[q * 4 + 2] += v.z; o[q * 4 + 3] += v.w;
                        }
            }
            if (pass == 0) __syncthreads();
        }
    }

    // ---- epilogue: C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int n = n0 + wn * 64 + 2 * l31;
    if (n >= p.N) return;
    float bias0 = 0.f, bias1 = 0.f;
    if (p.bias) {
        bias0 = DType<T>::to_f32(((const T*)p.bias)[n]);
        bias1 = DType<T>::to_f32(((const T*)p.bias)[n + 1]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (KG == 2 && (mt >> 1) != kg) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= p.M) continue;
            const T v0 = DType<T>::from_f32(acc[mt][0][r] + bias0), v1 = DType<T>::from_f32(acc[mt][1][r] + bias1);
            const unsigned o = (unsigned)__builtin_bit_cast(unsigned short, v0) | ((unsigned)__builtin_bit_cast(unsigned short, v1) << 16);
            *(unsigned*)((unsigned short*)p.out + (size_t)m * p.N + n) = o;
        }
    }
}

template <typename T, int BK, int KG, int ABL = 0>
hipError_t ldsb_launch_one(const LdsbParams& p, hipStream_t st, bool grant) {
    constexpr int lds = lb_lds_bytes<BK, KG>();
    auto* kern = gemm_ldsb_kernel<T, BK, KG, ABL>;
    if (grant) return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(p.nbm * p.nbn), dim3(256 * KG), lds, st, p);
    return hipGetLastError();
}
template <typename T>
hipError_t ldsb_launch_t(const LdsbParams& p, int bk, int kgroups, hipStream_t st, bool grant) {
    if (bk == 64 && kgroups == 1) return ldsb_launch_one<T, 64, 1>(p, st, grant);
    if (bk == 32 && kgroups == 1) return ldsb_launch_one<T, 32, 1>(p, st, grant);
    if (bk == 32 && kgroups == 2) return ldsb_launch_one<T, 32, 2>(p, st, grant);
    return hipErrorInvalidValue;
}

}  // namespace

bool ldsb_supported(const gptq_layer_t& L, int M) {
    const int spg = L.group_size / 32;
    return L.bits == 4 && (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16) && L.group_size % 32 == 0 && spg >= 1 && (spg & (spg - 1)) == 0 &&
           L.K % 64 == 0 && L.K % L.group_size == 0 && L.N % 8 == 0 && L.K / 64 >= 4 && M >= 1;
}

hipError_t init_gemm_ldsb_device() {
    LdsbParams p{};
    hipError_t e = hipSuccess;
    for (int bk : {64, 32})
        for (int kgr : {1, 2}) {
            if (bk == 64 && kgr == 2) continue;
            if (e == hipSuccess) e = ldsb_launch_t<f16>(p, bk, kgr, nullptr, true);
            if (e == hipSuccess) e = ldsb_launch_t<bf16>(p, bk, kgr, nullptr, true);
        }
    return e;
}

// x: [M, K] in the k order of `qweight` (for act-order layers: the permuted copy and qweight_seq).  bk / kgroups = 0: the planner's choice.
hipError_t launch_gemm_ldsb(const gptq_layer_t& L, const uint32_t* qweight, const void* x, void* out, int M, hipStream_t st, int bk, int kgroups, int abl) {
    LdsbParams p{};
    p.qweight = qweight; p.qzeros = L.qzeros; p.scales = L.scales; p.bias = L.bias;
    p.x = x; p.out = out;
    p.M = M; p.K = L.K; p.N = L.N; p.zero_mode = L.zero_mode;
    p.nbm = (M + LB_BM - 1) / LB_BM;
    p.nbn = (L.N + LB_BN - 1) / LB_BN;
    p.lg_spg32 = __builtin_ctz((unsigned)(L.group_size / 32));
    if (!bk) bk = 32;
    if (bk == 64 && L.group_size % 64) return hipErrorInvalidValue;
    if (!kgroups) kgroups = (p.nbm * p.nbn <= 256 && bk == 32 && (L.K / bk) % 4 == 0) ? 2 : 1;
    if (kgroups == 2 && (bk != 32 || (L.K / bk) % 4)) return hipErrorInvalidValue;      // an even number of steps per group (two per loop trip)
    p.ksteps = L.K / bk;
#ifdef GPTQ_LDSB_ABLATIONS
    if (abl) {
        if (L.dtype != GPTQ_F16 || bk != 32 || kgroups != 1) return hipErrorInvalidValue;
        hipError_t e;
        switch (abl) {
#define ABL_CASE(V) case V: e = ldsb_launch_one<f16, 32, 1, V>(p, st, true); return e != hipSuccess ? e : ldsb_launch_one<f16, 32, 1, V>(p, st, false);
            ABL_CASE(1) ABL_CASE(2) ABL_CASE(4) ABL_CASE(6) ABL_CASE(8) ABL_CASE(16) ABL_CASE(31)
#undef ABL_CASE
            default: return hipErrorInvalidValue;
        }
    }
#endif
    (void)abl;
    return L.dtype == GPTQ_BF16 ? ldsb_launch_t<bf16>(p, bk, kgroups, st, false) : ldsb_launch_t<f16>(p, bk, kgroups, st, false);
}

}  // namespace gptq
