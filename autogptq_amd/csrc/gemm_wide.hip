// gemm_wide.hip -- prefill, round 4: 128 x 512 workgroup tiles, every wave a 128 x 128 tile with its 256 accumulators in the AGPR half of the
// unified register file (one wave per SIMD, __launch_bounds__(256, 1) = 512 registers per lane).
//
// Replaces for LARGE launches (reference): Marlin<...> in autogptq_extension/marlin/marlin_cuda_kernel.cu:216-727 -- its goal, int4 x fp16 on the matrix
// cores at the dense rate -- and the dequant + cublasHgemm fallbacks (exllama/cuda_func/q4_matmul.cu:225-260, exllamav2/cuda/q_gemm.cu:104-181).
//
// Why a second tiled kernel: the counters of gemm_kernel<4, f16, 4, 64> (128 x 64 per wave, two waves per SIMD; profiles/r02_gemm_pmc.txt) show the
// matrix pipe 45 - 50 % busy with 5.7 VALU + 0.63 LDS per MFMA, and the round-3 ablation (profiles/r03_gemm_ldsb_lab.log) that the A path -- 16
// ds_read_b128, the x DMA and the barrier per 32 MFMAs -- caps that tile at ~0.55 of peak before a weight is decoded.  A 128 x 128 wave tile reads
// every A fragment once per 128 columns: 16 ds_read_b128, ONE barrier and half the x staging per 64 MFMAs; the dequantised B fragment is still reused by
// 4 row tiles (3.25 dequant VALU per MFMA: each wave decodes only its own 128 columns, nothing twice).
// Per wave and 64-deep K-step: 64 MFMAs (v_mfma_f32_32x32x16), 16 ds_read_b128, 4 x 16-byte weight loads + 2 group-constant loads, 4 x-tile chunks.
// A lane owns 4 ADJACENT columns (one 16-byte load per packed row; 512 contiguous bytes per half wave) -- column tile nt of the wave is the columns
// {4 l + nt}: the epilogue then stores 4 adjacent outputs (8 bytes) per lane and row, 256 contiguous bytes per half wave.
// Used where the launch has enough 128 x 512 tiles to fill 256 CUs (plan_gemm: north_star's M = 4096 on 4096 -> 4096 is exactly 256 of them); 128 x 256
// tiles (gemm.hip) stay for everything smaller -- M = 2048 on a 4096-wide layer is 128 wide tiles: half the chip.
// Weights never touch LDS; dequant = gemm.hip's exact magic-number form (bit-identical W, one rounding); x: LDS, double buffered, slot order
// (k0,k4,k1,k5,k2,k6,k3,k7) written by the staging pass (4 v_perm per 16 bytes) or, for act-order layers, delivered that way by the permute pre-pass and
// staged by LDS DMA with the XOR swizzle of gemm.hip.
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"
#include "launch.h"

namespace gptq {
namespace wide {

struct WideParams {
    const unsigned* qweight;      // TILED: the layer's decode copy (qweight_tiled), else the checkpoint / re-sequenced rows
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
    const void* x;
    void* out;
    int M, K, N, zero_mode, nbm, nbn, ksteps, qrows, groups, chunks;
    unsigned long long kpg_inv;   // ceil(2^32 / (group_size / 64)): group of K-step kt = (kt * kpg_inv) >> 32
};


// GLDS: x arrives in k-slot order (act-order layers: the permute pre-pass writes it so) and is staged by LDS DMA into unpadded, XOR-swizzled rows.
// TILED (implies GLDS): the weights come from the layer's DECODE COPY (gptq_prepack_decode: strips of 16 columns, a column's 32 consecutive k in 4
// adjacent words, nibbles in pair order).  The same extraction then yields (k, k + 1) pairs in the order x lies in memory, so the RAW x is DMA-staged: no
// register round trip, no v_perm, no ds_write (the ablation of the register-staged form put them at 9 % of the loop).  A lane still owns 4 adjacent
// columns: 4 x 16 bytes = 64 contiguous bytes per (lane, K-step); its half takes one 32-k slot of the step (k = 32 half + 8 ks for MFMA ks), and the A
// fragment of (ks, half) is piece 4 half + ks of the row's 128-byte step segment.
// G128: group_size % 128 == 0 -- the two 64-deep steps of a loop body lie in one group: constants are loaded and set up once per body, not per step.
template <typename T, bool GLDS, bool TILED = false, bool G128 = false>
__global__ void __launch_bounds__(256, 1) gemm_wide_kernel(WideParams p) {
    static_assert(!TILED || GLDS, "the decode copy is read with DMA-staged raw x");
    constexpr int BM = 128, BK = 64, KS = 4, MT = 4, NT = 4;
    constexpr int STRIDE = GLDS ? BK * 2 : BK * 2 + 16;      // bytes per LDS row of x (padded unless DMA-staged)
    constexpr int NCH = 4;                                   // 16-byte chunks per thread and K-step: 128 rows x 8 chunks / 256 threads
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x BM x STRIDE
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // logical tile order: an XCD's contiguous run of ids is a compact patch (all column tiles of a few row tiles): its L2 holds few x panels
    const int L = xcd_remap(blockIdx.x, p.nbm * p.nbn);
    const int bm = L / p.nbn, bn = L - bm * p.nbn;
    const int m0 = bm * BM;
    const int n = bn * 512 + wave * 128 + 4 * l31;            // this lane's first column (it owns n .. n + 3)
    const bool col_ok = n < p.N;                              // N % 32 == 0: a lane's 4 columns are in or out together
    const int nl = col_ok ? n : 0;
    const unsigned short* __restrict__ x = (const unsigned short*)p.x;

    // A staging assignment: chunk c -> (row, 16-byte column); rows past M are clamped (never predicated), their accumulators are not stored
    int a_row[NCH], a_kc[NCH];
    unsigned a_off[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * 256;
        a_row[i] = c >> 3;
        a_kc[i] = c & 7;
        const int src_kc = GLDS ? (a_kc[i] ^ ((a_row[i] >> 1) & 7)) : a_kc[i];
        a_off[i] = (unsigned)(min(m0 + a_row[i], p.M - 1) - m0) * (unsigned)p.K * 2u + (unsigned)src_kc * 16u;
    }
    const char* a_base = (const char*)(x + (size_t)m0 * p.K);
    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qweight, 0, TILED ? (int)((size_t)(p.N / 16) * p.chunks * 1024) : (int)((size_t)p.qrows * p.N * 4), 0x00020000);
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)((size_t)BM * p.K * 2), 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, 0, (int)((size_t)p.groups * p.N * 2), 0x00020000);
    const int zrow_bytes = p.N / 8 * 4;
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.qzeros, 0, p.groups * zrow_bytes, 0x00020000);
    const unsigned b_lane_off = TILED ? ((unsigned)nl >> 4) * (unsigned)p.chunks * 1024u + (unsigned)half * 256u + ((unsigned)nl & 15u) * 16u    // strip, k-slot, column
                                      : ((unsigned)nl + (unsigned)half * (unsigned)p.N) * 4u;
    const unsigned s_lane_off = (unsigned)nl * 2u;
    const unsigned z_lane_off = ((unsigned)nl >> 3) * 4u, zsh = ((unsigned)nl & 7u) * 4u;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;

    auto dma_a = [&](int kt, int buf) {
        // the four DMAs of a thread as ONE asm block: m0 saved / restored once, the step's base in an SGPR pair, the lane's four offsets fixed in VGPRs (no
        // per-step address VALU; -1 us of 119 at M = 4096 on 4096^2 against four lds_dma16 calls, tools/widelab2)
        const char* sb = a_base + (size_t)kt * (BK * 2);
        const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + (size_t)buf * (BM * STRIDE) + (size_t)(wave * 64) * 16));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %9\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %9\n\t"
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %9\n\t"
                     "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %9\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(a_off[0]), "v"(a_off[1]), "v"(a_off[2]), "v"(a_off[3]), "s"(l0), "s"(l0 + 4096u), "s"(l0 + 8192u), "s"(l0 + 12288u), "s"(sb)
                     : "memory");
    };
    auto load_a = [&](int kt, u32x4 (&r)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, a_off[i], (unsigned)kt * (BK * 2), 0);
    };
    auto store_a = [&](int buf, const u32x4 (&r)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            // (x0,x1)(x2,x3)(x4,x5)(x6,x7) -> (x0,x4)(x1,x5)(x2,x6)(x3,x7): the slot order of the B fragments
            u32x4 o;
            o[0] = __builtin_amdgcn_perm(r[i][2], r[i][0], 0x05040100u);
            o[1] = __builtin_amdgcn_perm(r[i][2], r[i][0], 0x07060302u);
            o[2] = __builtin_amdgcn_perm(r[i][3], r[i][1], 0x05040100u);
            o[3] = __builtin_amdgcn_perm(r[i][3], r[i][1], 0x07060302u);
            *(u32x4*)(smem + (size_t)buf * (BM * STRIDE) + a_row[i] * STRIDE + a_kc[i] * 16) = o;
        }
    };
    // weights of one K-step: packed rows kt * 8 + ks * 2 + half, 16 bytes (4 columns) each
    // TILED: b[col] = the 4 words (MFMA steps ks = 0..3) of column n + col in k-slot 2 (kt & 1) + half of chunk kt / 2
    auto load_b = [&](int kt, u32x4 (&b)[KS]) {
        if constexpr (TILED) {
            const unsigned so = (unsigned)(kt >> 1) * 1024u + (unsigned)(kt & 1) * 512u;
#pragma unroll
            for (int col = 0; col < NT; ++col) b[col] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, b_lane_off + col * 16u, so, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                b[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, b_lane_off, (unsigned)((size_t)(kt * 8 + ks * 2) * (size_t)p.N * 4), 0);
        }
    };
    auto bw = [](const u32x4 (&b)[KS], int ks, int nt) -> unsigned { return TILED ? b[nt][ks] : b[ks][nt]; };      // the word of MFMA step ks, column tile nt
    auto a_piece = [&](int ks) -> int { return TILED ? (half * 4 + ks) : (ks * 2 + half); };                          // 16-byte piece of the row's step segment
    auto load_c = [&](int kt, CRaw& c) {
        const int g = (int)(((unsigned long long)(unsigned)kt * p.kpg_inv) >> 32);
        c.s = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_s, s_lane_off, (unsigned)g * (unsigned)p.N * 2u, 0));
        c.z = __builtin_amdgcn_raw_buffer_load_b32(rsrc_z, z_lane_off, (unsigned)(g * zrow_bytes), 0);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    u32x4 a_next[NCH];
    u32x4 b0[KS], b1[KS];
    CRaw c0, c1;
    const int kt1 = p.ksteps;
    if constexpr (GLDS) dma_a(0, 0); else load_a(0, a_next);
    load_b(0, b0);
    load_c(0, c0);
    if constexpr (!GLDS) store_a(0, a_next);
    if constexpr (GLDS) wait_vmcnt<0>();
    __syncthreads();

    const int a_lane_off = GLDS ? l31 * STRIDE : l31 * STRIDE + half * 16;
    const int a_swz = (l31 >> 1) & 7;
    // One wave per SIMD: nothing else covers this wave's VALU / LDS work, so every MFMA has to be followed by its share of it -- the matrix pipe takes an
    // MFMA every 32 cycles and the wave can issue ~5 other instructions meanwhile (MI355X_MICROARCH.md, per-instruction table).  Left to itself hipcc emits
    // the 52-instruction dequant of a k-step as one run and the 16 MFMAs as another (the pipe idles through the first and the wave stalls through the
    // second: 51 % busy, 1128 TFLOP/s at 4096^3 in tools/widelab); sched_group_barrier spells the interleave out.  The pipeline is continuous ACROSS
    // K-steps: the constants and the first B fragments of step t + 1 are produced under the last 16 MFMAs of step t (their words were requested a whole
    // step earlier), and the x tile of t + 1 goes to LDS there too; only the first A fragments of a step wait for the barrier.
    Deq4<T> dq_cur;
    u32x4 bq_first[NT];
    dq_cur.setup(c0, zsh, zmask);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bq_first[nt] = dq_cur.frag(bw(b0, 0, nt), nt);
    auto interleave = [&](auto nvalu) {                       // 16 x { 1 MFMA, n VALU, 1 LDS op every fourth }
        constexpr int NV = decltype(nvalu)::value;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);       // VALU
            if ((i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);   // DS read / write
        }
    };
    auto step = [&](int kt, auto bufc, const u32x4 (&b_use)[KS], u32x4 (&b_fill)[KS], CRaw& c_fill) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool NEWG = !(G128 && BUF == 0);             // does step kt + 1 open a new group?  (G128: only behind the odd step of a body)
        const int ktn = min(kt + 1, kt1 - 1);                  // the last step re-loads itself (no branch in the pipeline)
        if constexpr (GLDS) {
            // claim this step's weight words before anything new is issued: the compiler's exact wait lands here (gemm.hip)
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) asm volatile("" ::"v"(b_use[ks][0]), "v"(b_use[ks][1]), "v"(b_use[ks][2]), "v"(b_use[ks][3]));
            __builtin_amdgcn_sched_barrier(0);
        } else {
            load_a(ktn, a_next);
        }
        // the next step's weight / constant loads and its x DMA are issued BETWEEN this step's MFMA groups (below), not here: in front of the groups they
        // were ~50 instructions the matrix pipe waited through (-2 us of 119, tools/widelab2: the loads' cost is their issue, not their latency --
        // replacing every step's addresses by step 0's, cache-resident, changed 1 us; removing the loads 6)
        const char* abase = smem + BUF * (BM * STRIDE) + a_lane_off;
        u32x4 a[2][MT], bq[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[0][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (GLDS ? ((a_piece(0) ^ a_swz) * 16) : 0));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[0][nt] = bq_first[nt];
        __builtin_amdgcn_sched_barrier(0);                     // the prefetch and the first A fragments stay ahead of this step's MFMAs
        Deq4<T> dq_nx;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#if defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[(ks + 1) & 1][mt] = a[ks & 1][mt] ^ u32x4{1u, 2u, 3u, (unsigned)ks};   // lab: no A-fragment LDS reads after the first
#else
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[(ks + 1) & 1][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (GLDS ? ((a_piece(ks + 1) ^ a_swz) * 16) : (ks + 1) * 32));
#endif
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[(ks + 1) & 1][nt] = dq_cur.frag(bw(b_use, ks + 1, nt), nt);
            } else {
                // under the last MFMA group: next step's constants and first B fragments (registers only), and its x tile to LDS
                if constexpr (NEWG) dq_nx.setup(c_fill, zsh, zmask);
                else dq_nx = dq_cur;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq_first[nt] = dq_nx.frag(bw(b_fill, 0, nt), nt);
#if !(defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 8))
                if constexpr (!GLDS) store_a(BUF ^ 1, a_next);
#endif
            }
            if (ks == 0) {                                     // next step's weights + constants: under MFMA group 0
#if defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 32)
#pragma unroll
                for (int i = 0; i < KS; ++i) b_fill[i] = b_use[i] ^ u32x4{1u, 2u, 3u, 4u};      // lab: no weight loads after the first step
#elif defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 64)
                load_b(0, b_fill);                             // lab: always step 0's (cache-resident) words: issue cost without the latency
#else
                load_b(ktn, b_fill);
#endif
                if constexpr (NEWG) load_c(ktn, c_fill);
            }
            if constexpr (GLDS) if (ks == 1) {                 // next step's x tile: under group 1
#if defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 128)
                dma_a(0, BUF ^ 1);
#elif !(defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 16))
                dma_a(ktn, BUF ^ 1);
#endif
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Mma<T>::run(a[ks & 1][mt], bq[ks & 1][nt], acc[mt][nt]);
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if ((i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);
                    if ((i & 1) == 1 && i < 12) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 6 VMEM reads: the 4 weight + 2 constant loads
                }
            } else if (ks + 1 < KS) interleave(std::integral_constant<int, 4>{});
            else interleave(std::integral_constant<int, 6>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        dq_cur = dq_nx;
        // DMA-staged x: the next tile must have landed before anybody passes the barrier
        if constexpr (GLDS) wait_vmcnt<0>();                   // the DMAs are the step's newest VMEM operations
#if !(defined(GPTQ_WIDE_ABL) && (GPTQ_WIDE_ABL & 4))
        __syncthreads();
#endif
    };
    for (int kt = 0; kt < kt1; kt += 2) {                       // the planner only sends even step counts here (K % 128 == 0): no conditional second step --
        step(kt, std::integral_constant<int, 0>{}, b0, b1, c1);         // with all 256 accumulator registers live, a phi copy of them has nowhere to go but scratch
        step(kt + 1, std::integral_constant<int, 1>{}, b1, b0, c0);
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: column = lane & 31 (of tile nt: output column n + nt), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (!col_ok) return;
    float bias[NT] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bias[nt] = DType<T>::to_f32(((const T*)p.bias)[n + nt]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= p.M) continue;
            const unsigned lo = (unsigned)t_bits(DType<T>::from_f32(acc[mt][0][r] + bias[0])) | ((unsigned)t_bits(DType<T>::from_f32(acc[mt][1][r] + bias[1])) << 16);
            const unsigned hi = (unsigned)t_bits(DType<T>::from_f32(acc[mt][2][r] + bias[2])) | ((unsigned)t_bits(DType<T>::from_f32(acc[mt][3][r] + bias[3])) << 16);
            *(u32x2*)((unsigned short*)p.out + (size_t)m * p.N + n) = u32x2{lo, hi};
        }
    }
}

}  // namespace wide

// ---- host side -------------------------------------------------------------------------------------------------------------------------------
bool wide_gemm_ok(const gptq_layer_t& L, int M, bool use_seq, bool xslot_glds) {
    if (L.bits != 4 || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16)) return false;
    if (L.K % 128 || L.group_size % 64 || L.N % 32 || L.epilogue != GPTQ_EPI_NONE) return false;      // an even number of 64-deep K-steps
    if (use_seq && !xslot_glds) return false;                 // act-order: only with the slot-ordered, DMA-staged x (fp16)
    (void)M;
    return true;
}

hipError_t launch_gemm_wide(const gptq_layer_t& L, const uint32_t* qweight, const void* x, void* out, int M, bool glds, hipStream_t st, bool tiled) {
    wide::WideParams p{};
    p.qweight = tiled ? L.qweight_tiled : qweight;
    p.chunks = L.K / 128; p.qzeros = L.qzeros; p.scales = L.scales; p.bias = L.bias; p.x = x; p.out = out;
    p.M = M; p.K = L.K; p.N = L.N; p.zero_mode = L.zero_mode;
    p.nbm = (M + 127) / 128; p.nbn = (L.N + 511) / 512;
    p.ksteps = L.K / 64;
    p.qrows = L.K / 8;
    p.groups = (L.K + L.group_size - 1) / L.group_size;
    const unsigned long long kpg = (unsigned long long)(L.group_size / 64);
    p.kpg_inv = ((1ull << 32) + kpg - 1) / kpg;
    const dim3 grid(p.nbm * p.nbn), block(256);
    if (tiled) {                                              // plain layer with its decode copy: raw x by DMA
        const bool g128 = L.group_size % 128 == 0;
        if (L.dtype == GPTQ_F16) {
            if (g128) hipLaunchKernelGGL((wide::gemm_wide_kernel<f16, true, true, true>), grid, block, 2 * 128 * 128, st, p);
            else hipLaunchKernelGGL((wide::gemm_wide_kernel<f16, true, true, false>), grid, block, 2 * 128 * 128, st, p);
        } else {
            if (g128) hipLaunchKernelGGL((wide::gemm_wide_kernel<bf16, true, true, true>), grid, block, 2 * 128 * 128, st, p);
            else hipLaunchKernelGGL((wide::gemm_wide_kernel<bf16, true, true, false>), grid, block, 2 * 128 * 128, st, p);
        }
        return hipGetLastError();
    }
    if (L.dtype == GPTQ_F16) {
        if (glds) hipLaunchKernelGGL((wide::gemm_wide_kernel<f16, true>), grid, block, 2 * 128 * 128, st, p);
        else hipLaunchKernelGGL((wide::gemm_wide_kernel<f16, false>), grid, block, 2 * 128 * 144, st, p);
    } else {
        hipLaunchKernelGGL((wide::gemm_wide_kernel<bf16, false>), grid, block, 2 * 128 * 144, st, p);
    }
    return hipGetLastError();
}

}  // namespace gptq
