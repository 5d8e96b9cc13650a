// gemm_wide_sk_kernel.cuh -- the stream-K prefill kernel's body (see gemm_wide_sk.hip for the schedule), written once for the three packings of the decode copy:
//   BITS = 4: gemm_wide_sk.hip (gemm_wide_sk_kernel<T, G128>: constants from the checkpoint rows, as gemm_wide.hip reads them)
//   BITS = 3 / 8 and the 32-wide groups of BASELINE config 5: gemm_wide_sk_b38.hip (gemm_wide_skb_kernel<T, BITS, GM>)
// GM = the group mode: 0 = group_size % 128 == 0 (new constants every second 64-deep step), 1 = group_size % 64 == 0, 2 = group_size == 32.  A lane of the
// 32x32x16 matrix-core step holds k = 32 half + 8 ks + 0..7 of the step (both operands: the x tile is read from LDS in that order), so with 32-wide groups
// each half of the wave simply owns ITS group's constants -- no extra instruction in the loop.
#pragma once
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"

namespace gptq {
namespace wide {

struct WskParams {
    const unsigned* qweight;      // the layer's decode copy (qweight_tiled)
    const void* qconst;           // 3 / 8 bits: its constant records (qconst_tiled); 4 bits reads qzeros / scales below
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
    const void* x;
    void* out;
    int M, K, N, zero_mode, nbm, nbn, groups, chunks;
    int upt;                      // units per tile = K / 256
    int units_total;              // nbm * nbn * upt
    int lg_nwg;                   // the grid is 2^lg_nwg workgroups
    unsigned max_spins;
    unsigned long long kpg_inv;   // ceil(2^32 / (group_size / 64)): group of K-step kt = (kt * kpg_inv) >> 32
    unsigned* flags;              // workspace header: [workgroup] "my piece is published"; zero before and after every launch
    float* slots;                 // [workgroup][wave][32 quads][64 lanes] float4: the 64-row sums (both K parts) of a published piece
    unsigned* err;                // sticky error word: a bounded wait gave up
};

constexpr int WSK_XT_BYTES = 128 * 128;                       // one x tile: 128 rows x 64 k x 2 bytes
constexpr int WSK_EX_OFFSET = 4 * WSK_XT_BYTES;               // the four x tiles (2 buffers x 2 K parts x 16 KiB) ...
constexpr int WSK_LDS_BYTES = WSK_EX_OFFSET + 4 * 16384;      // ... and the exchange area behind them (4 waves x 16 KiB per pass)
constexpr size_t WSK_WAVE_SLOT_FLOATS = (size_t)32 * 64 * 4;  // a wave's half of its 128 x 128 tile, the two K parts summed: 32 KiB
constexpr size_t WSK_SLOT_FLOATS = 4 * WSK_WAVE_SLOT_FLOATS;   // one workgroup's published piece: 128 KiB

// a lane's packed weights of one 64-deep step, 4 columns: 4 bits: word ks of column col = w[col][ks]; 3 bits: the lane's three words per column
template <int BITS> struct WRaw { u32x4 w[4]; };
template <> struct WRaw<3> { u32x3 w[4]; };
template <typename T, int BITS> struct DeqOf { typedef Deq4<T> type; };
template <typename T> struct DeqOf<T, 3> { typedef Deq3<T> type; };
template <typename T> struct DeqOf<T, 8> { typedef Deq8<T> type; };
// 8 bits: a step's weights are 8 words per column -- twice the registers of the 4-bit form, which the kernel does not have (all 256 accumulator registers
// and ~250 of the 256 others are live).  Three 4-word sets rotate instead of two 8-word buffers: a step reads LO (MFMA steps 0, 1) and HI (2, 3); the next
// step's LO words are loaded into the FREE set under MFMA group 0 and its HI words into LO under group 1, when LO's last words have been dequantised.
// Then (lo, hi, fr) <- (fr, lo, hi).
struct Use8 { WRaw<8>& lo; WRaw<8>& hi; WRaw<8>& fr; };

template <typename T, int BITS, int GM>
__device__ __forceinline__ void wsk_body(const WskParams& p) {
    constexpr int KS = 4, MT = 4, NT = 4, STRIDE = 128;
    constexpr bool G128 = GM == 0, G32 = GM == 2;
    constexpr unsigned CHUNK_BYTES = BITS == 3 ? 768u : 1024u;          // one (strip, 128-deep chunk) of the decode copy; 8 bits: 64-deep chunks
    constexpr unsigned REC = BITS == 8 ? 64u : 48u;                     // one (strip, group) constant record
    using CR = std::conditional_t<BITS == 8, CRaw8, CRaw>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, kp = wave >> 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int nwg = 1 << p.lg_nwg;
    const int Lb = xcd_remap(blockIdx.x, nwg);                 // consecutive ranges on one XCD: the pieces of a cut tile and a row tile's x panel share an L2
    auto range_start = [&](int b) -> int { return (int)(((unsigned long long)(unsigned)b * (unsigned long long)(unsigned)p.units_total) >> p.lg_nwg); };
    int u0 = range_start(Lb);
    const int u1 = range_start(Lb + 1);

    // x-tile DMA of a wave: 8 instructions, instruction i fills LDS rows 8 (8 cw + i) .. + 7 of the wave's K part (64 lanes x 16 bytes = 8 rows of 128 bytes);
    // LDS slot s of row R holds piece s ^ ((R >> 1) & 7) of the row's step segment (the swizzle of gemm.hip): (R >> 1) & 7 = 4 (i & 1) + (lane >> 4)
    unsigned a_voff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const unsigned r8 = (unsigned)lane >> 3, kc = (unsigned)lane & 7u;
        a_voff[par] = r8 * (unsigned)p.K * 2u + ((kc ^ (4u * par + (r8 >> 1))) * 16u);
    }
    const size_t a_grp_bytes = (size_t)8 * p.K * 2;            // 8 rows of x
    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qweight, 0, (int)((size_t)(p.N / 16) * p.chunks * CHUNK_BYTES), 0x00020000);
    // constants: 4 bits from the checkpoint rows (scales [G][N], qzeros [G][N / 8]), 3 bits from the decode copy's records [strip][G][48 bytes]
    const auto rsrc_s = BITS == 4 ? __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, 0, (int)((size_t)p.groups * p.N * 2), 0x00020000)
                                  : __builtin_amdgcn_make_buffer_rsrc((void*)p.qconst, 0, (int)((size_t)(p.N / 16) * p.groups * REC), 0x00020000);
    const int zrow_bytes = p.N / 8 * 4;
    const auto rsrc_z = BITS == 4 ? __builtin_amdgcn_make_buffer_rsrc((void*)p.qzeros, 0, p.groups * zrow_bytes, 0x00020000) : rsrc_s;
    const unsigned srow_bytes = BITS == 4 ? (unsigned)p.N * 2u : REC;  // from one group's constants to the next: scales, zero-points
    const unsigned zrow_step = BITS == 4 ? (unsigned)zrow_bytes : REC;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;
    const int a_lane_off = l31 * STRIDE;
    const int a_swz = (l31 >> 1) & 7;

    // per-tile state (set at the top of every segment)
    const char* a_tile = nullptr;                              // x + m0 * K
    unsigned b_lane_off = 0, s_lane_off = 0, z_lane_off = 0, zsh = 0;
    int kt_last = 0;

    auto dma_a4 = [&](int kt, int buf, int q) {                // DMAs 4 q .. 4 q + 3 of the wave's eight
#if defined(GPTQ_WSK_ABL) && (GPTQ_WSK_ABL & 4)
        if (kt != 0x7fffffff) return;                          // lab: no x DMAs (the tiles keep whatever the LDS held)
#endif
        // LDS rows 8 j .. 8 j + 7 (j = 8 cw + 4 q + i) take the rows of 32-row block j / 4 -- of block 3 - j / 4 for K part 1 (finish() relies on it)
        const int j0 = cw * 8 + q * 4;
        const char* sb = a_tile + (size_t)kt * 128 + (size_t)((kp ? 3 - (j0 >> 2) : (j0 >> 2)) * 4) * a_grp_bytes;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + (size_t)(buf * 2 + kp) * WSK_XT_BYTES + (size_t)(cw * 8 + q * 4) * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %7\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %8\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %9\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %10\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(a_voff[0]), "v"(a_voff[1]), "s"(l0), "s"(l0 + 1024u), "s"(l0 + 2048u), "s"(l0 + 3072u), "s"(sb), "s"(sb + a_grp_bytes),
                       "s"(sb + 2 * a_grp_bytes), "s"(sb + 3 * a_grp_bytes)
                     : "memory");
    };
    // b[col] = the 4 words (MFMA steps ks = 0..3) of column n + col in k-slot 2 (kt & 1) + half of chunk kt / 2
    auto load_b = [&](int kt, WRaw<BITS>& b) {
        const unsigned so = (unsigned)(kt >> 1) * CHUNK_BYTES + (unsigned)(kt & 1) * (CHUNK_BYTES / 2);
#pragma unroll
        for (int col = 0; col < NT; ++col) {
            if constexpr (BITS == 3) b.w[col] = __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(rsrc_q, b_lane_off + col * 12u, so, 0));
            else b.w[col] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, b_lane_off + col * 16u, so, 0);
        }
    };
    auto load_b8 = [&](int kt, WRaw<8>& b, unsigned hi) {      // 8 bits: k-slots 2 half + hi of chunk kt (16 k each): words 4 hi .. 4 hi + 3 of the lane's eight
#pragma unroll
        for (int col = 0; col < NT; ++col) b.w[col] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, b_lane_off + hi * 256u + col * 16u, (unsigned)kt * 1024u, 0);
    };
    auto load_c = [&](int kt, CR& c) {
        // 32-wide groups: the lane's group is 2 kt + half, and half is part of s_lane_off / z_lane_off
        const int g = G32 ? 2 * kt : (int)(((unsigned long long)(unsigned)kt * p.kpg_inv) >> 32);
        c.s = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_s, s_lane_off, (unsigned)g * srow_bytes, 0));
        if constexpr (BITS == 8) c.z = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_z, z_lane_off, (unsigned)g * zrow_step, 0));
        else c.z = __builtin_amdgcn_raw_buffer_load_b32(rsrc_z, z_lane_off, (unsigned)g * zrow_step, 0);
    };
    auto setup_dq = [&](typename DeqOf<T, BITS>::type& d, const CR& c) {
        if constexpr (BITS == 4) d.setup(c, zsh, zmask);
        else d.setup(c);
    };
    auto frag_of = [&](const typename DeqOf<T, BITS>::type& d, const auto& b, int nt, int ks) -> u32x4 {
#if defined(GPTQ_WSK_ABL) && (GPTQ_WSK_ABL & 8)
        if constexpr (BITS == 4) { const unsigned q = b.w[nt][ks]; return u32x4{q, q ^ 0x11111111u, q ^ 0x22222222u, q ^ f16x2_bits(d.c1[nt])}; }      // lab: no dequant math
#endif
        if constexpr (BITS == 4) return d.frag(b.w[nt][ks], nt);
        else if constexpr (BITS == 3) return d.frag(b.w[nt], ks, nt);
        else return ks < 2 ? d.frag(b.lo.w[nt][2 * ks], b.lo.w[nt][2 * ks + 1], nt) : d.frag(b.hi.w[nt][2 * ks - 4], b.hi.w[nt][2 * ks - 3], nt);
    };

    f32x16 acc[MT][NT];
    WRaw<BITS> b0, b1, b2;                                    // (b2: 8 bits only)
    CR c0, c1;
    typename DeqOf<T, BITS>::type dq_cur;
    u32x4 bq_first[NT];

    auto interleave = [&](auto nvalu) {                       // 16 x { 1 MFMA, n VALU, 1 LDS op every fourth }
        constexpr int NV = decltype(nvalu)::value;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);       // VALU
            if ((i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);   // DS read
        }
    };
    // One 64-deep K-step of this wave's K part (gemm_wide.hip's pipeline: the next step's constants and first B fragments under the last MFMA group, its
    // weight / constant loads inside group 0, its x DMAs inside groups 1 and 2).
    auto step = [&](int kt, auto bufc, const auto& b_use, auto& b_fill, CR& c_fill) {      // 8 bits: b_use = b_fill = the Use8 of the step
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool NEWG = !(G128 && BUF == 0);             // does step kt + 1 open a new group?  (G128: only behind the odd step of a body)
        const int ktn = min(kt + 1, kt_last);                  // the segment's last step re-loads itself (no branch in the pipeline)
        // claim this step's weight words before anything new is issued: the compiler's exact wait lands here (gemm.hip)
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
            if constexpr (BITS == 3) asm volatile("" ::"v"(b_use.w[ks][0]), "v"(b_use.w[ks][1]), "v"(b_use.w[ks][2]));
            else if constexpr (BITS == 8) asm volatile("" ::"v"(b_use.lo.w[ks][0]), "v"(b_use.lo.w[ks][1]), "v"(b_use.lo.w[ks][2]), "v"(b_use.lo.w[ks][3]),
                                                       "v"(b_use.hi.w[ks][0]), "v"(b_use.hi.w[ks][1]), "v"(b_use.hi.w[ks][2]), "v"(b_use.hi.w[ks][3]));
            else asm volatile("" ::"v"(b_use.w[ks][0]), "v"(b_use.w[ks][1]), "v"(b_use.w[ks][2]), "v"(b_use.w[ks][3]));
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* abase = smem + (size_t)(BUF * 2 + kp) * WSK_XT_BYTES + a_lane_off;
        u32x4 a[2][MT], bq[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[0][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (((half * 4 + 0) ^ a_swz) * 16));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[0][nt] = bq_first[nt];
        __builtin_amdgcn_sched_barrier(0);
        typename DeqOf<T, BITS>::type dq_nx;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[(ks + 1) & 1][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (((half * 4 + ks + 1) ^ a_swz) * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[(ks + 1) & 1][nt] = frag_of(dq_cur, b_use, nt, ks + 1);
            } else {
                if constexpr (NEWG) setup_dq(dq_nx, c_fill);
                else dq_nx = dq_cur;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (BITS == 8) bq_first[nt] = dq_nx.frag(b_fill.fr.w[nt][0], b_fill.fr.w[nt][1], nt);
                    else bq_first[nt] = frag_of(dq_nx, b_fill, nt, 0);
                }
            }
            if (ks == 0) {                                     // next step's weights + constants: under MFMA group 0
                if constexpr (BITS == 8) load_b8(ktn, b_fill.fr, 0u);
                else load_b(ktn, b_fill);
                if constexpr (NEWG) load_c(ktn, c_fill);
                dma_a4(ktn, BUF ^ 1, 0);                       // the next step's x tile: behind the weight loads and inside group 1 (in groups 1 and 2 the
            }                                                  // second half lands too late: the compiler's wait for the weights, in front of group 3's
            if (ks == 1) {                                     // VALU work, is a vmcnt(0) that covers the DMAs as well: +5 % measured)
                dma_a4(ktn, BUF ^ 1, 1);
                if constexpr (BITS == 8) load_b8(ktn, b_fill.lo, 1u);      // LO's words 2, 3 went into bq[1] under group 0
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
            } else if (BITS == 8 && ks == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if ((i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);
                    if ((i & 1) == 1 && i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // the 4 HI loads
                }
            } else if (ks + 1 < KS) interleave(std::integral_constant<int, 4>{});
            else interleave(std::integral_constant<int, 6>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        dq_cur = dq_nx;
#if defined(GPTQ_WSK_ABL) && (GPTQ_WSK_ABL & 1)
        // lab (tools/lab/wsk_ablate.sh; wrong results by construction): nobody waits for the step's loads
#else
        wait_vmcnt<0>();                                       // the next tile must have landed before anybody passes the barrier (the DMAs are the step's newest VMEM operations)
#endif
#if defined(GPTQ_WSK_ABL) && (GPTQ_WSK_ABL & 2)
        // lab: no barrier between the steps
#else
        __syncthreads();
#endif
    };

    // End of a segment.  K part 1 keeps its x tile with the four 32-row blocks in REVERSED order (dma_a4), so its accumulators acc[mt] belong to row block
    // 3 - mt: in register terms every wave keeps acc[0..1] and hands acc[2..3] to its partner (same columns, other K part) through LDS -- one code path
    // for both K parts (a branch over which accumulators to read makes hipcc copy all 256 out of the AGPRs behind the K loop and spill them).  Two passes
    // of 16 KiB per wave (acc[3] -> the partner's acc[0], then acc[2] -> its acc[1]): the exchange area does not alias the x tiles, whose first buffer
    // is already being filled for the next segment.
    // pub: the piece is a later part of a tile another workgroup finishes -- the 64-row sums of the two K parts go to this workgroup's slot (write-through
    // stores, 32 KiB per wave).  Else the published pieces of the tile (workgroups Lb + 1 .. Lb + nb, in range order) are added and the outputs stored.
    auto finish = [&](bool pub, int m0, int m_lo, int bn, int lane_e, int nb) {
        const int half_e = lane_e >> 5;
        const int n = bn * 256 + cw * 128 + 4 * (lane_e & 31);
        const bool col_ok = n < p.N;
        char* const ex_mine = smem + WSK_EX_OFFSET + (size_t)wave * 16384 + (size_t)lane_e * 16;
        const char* const ex_partner = smem + WSK_EX_OFFSET + (size_t)(wave ^ 2) * 16384 + (size_t)lane_e * 16;
        float bias[NT] = {0.f, 0.f, 0.f, 0.f};
        if (!pub && p.bias && col_ok) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bias[nt] = DType<T>::to_f32(((const T*)p.bias)[n + nt]);
        }
        float* const my_slot = p.slots + ((size_t)Lb * 4 + wave) * WSK_WAVE_SLOT_FLOATS + (size_t)lane_e * 4;
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl) {
            const int rb = kp ? 3 - mtl : mtl;                 // the 32-row block acc[mtl] holds
            if (mtl == 1) __syncthreads();                     // pass 0's values have been read
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16& a = acc[3 - mtl][nt];        // the partner's acc[mtl] is the same row block
                    const f32x4 v = {a[rq * 4], a[rq * 4 + 1], a[rq * 4 + 2], a[rq * 4 + 3]};
                    *(f32x4*)(ex_mine + (size_t)((nt * 4 + rq) * 1024)) = v;
                }
            __syncthreads();                                   // ... and (pass 0) the finisher's flag waits are behind everybody
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {                   // two row quads (8 rows x 4 columns per lane) at a time
                f32x4 v[2][NT];
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int rq = rp * 2 + r2;
                        const f32x16& a = acc[mtl][nt];
                        const f32x4 mine = {a[rq * 4], a[rq * 4 + 1], a[rq * 4 + 2], a[rq * 4 + 3]};
                        const f32x4 theirs = *(const f32x4*)(ex_partner + (size_t)((nt * 4 + rq) * 1024));
                        v[r2][nt] = mine + theirs;             // fp32 a + b == b + a: the same bits whichever K part this wave is
                    }
                if (pub) {
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            // s_nop inside the string: nothing is padded behind an asm statement, and the next instruction may overwrite the data registers
                            // (dead to the compiler) while the store still reads them (gemm.hip, tools/tail_diag.py)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(my_slot + (size_t)(((mtl * 4 + rp * 2 + r2) * NT + nt) * 256)), "v"(v[r2][nt]) : "memory");
                        }
                } else {
                    for (int s = 0; s < nb; ++s) {             // published pieces, in range order: 8 x 16-byte loads in flight per lane
                        const float* slot = p.slots + ((size_t)(Lb + 1 + s) * 4 + wave) * WSK_WAVE_SLOT_FLOATS + (size_t)lane_e * 4;
                        unsigned long long w[2][NT][2];
#pragma unroll
                        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {  // agent-scope loads: they bypass this XCD's non-coherent L2 lines
                                const unsigned long long* src = (const unsigned long long*)(slot + (size_t)(((mtl * 4 + rp * 2 + r2) * NT + nt) * 256));
                                w[r2][nt][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                w[r2][nt][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
#pragma unroll
                        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                v[r2][nt][0] += __builtin_bit_cast(float, (unsigned)w[r2][nt][0]);
                                v[r2][nt][1] += __builtin_bit_cast(float, (unsigned)(w[r2][nt][0] >> 32));
                                v[r2][nt][2] += __builtin_bit_cast(float, (unsigned)w[r2][nt][1]);
                                v[r2][nt][3] += __builtin_bit_cast(float, (unsigned)(w[r2][nt][1] >> 32));
                            }
                    }
                    if (col_ok) {
#pragma unroll
                        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {      // C/D layout of the 32x32 MFMA: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), r = 4 rq + i
                                const int m = m0 + rb * 32 + i + 8 * (rp * 2 + r2) + 4 * half_e;
                                if (m < m_lo) continue;        // the shifted last row tile stores only its own rows (the rows above belong to the tile before it)
                                const unsigned lo = (unsigned)t_bits(DType<T>::from_f32(v[r2][0][i] + bias[0])) | ((unsigned)t_bits(DType<T>::from_f32(v[r2][1][i] + bias[1])) << 16);
                                const unsigned hi = (unsigned)t_bits(DType<T>::from_f32(v[r2][2][i] + bias[2])) | ((unsigned)t_bits(DType<T>::from_f32(v[r2][3][i] + bias[3])) << 16);
                                *(u32x2*)((unsigned short*)p.out + (size_t)m * p.N + n) = u32x2{lo, hi};
                            }
                    }
                }
            }
        }
        if (pub) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every publishing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.flags + Lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // Segments: the pieces of tiles inside [u0, u1).  open_segment() sets the per-tile state of the segment starting at u0 and ISSUES its first loads (x tile
    // of the first step into buffer 0, weights and constants into b0 / c0); it is called for segment s + 1 between the K loop and the epilogue of segment s,
    // so that the memory latency of a segment's first step lies under the previous segment's exchange and stores.
    int t = 0, j0 = 0, len = 0, bm = 0, bn = 0, m0 = 0, kt0 = 0, kt1 = 0;
    auto open_segment = [&]() {
        t = u0 / p.upt;
        j0 = u0 - t * p.upt;
        len = min(p.upt - j0, u1 - u0);
        bm = t / p.nbn;
        bn = t - bm * p.nbn;
        m0 = min(bm * 128, p.M - 128);
        const int n = bn * 256 + cw * 128 + 4 * l31;          // this lane's first column (it owns n .. n + 3)
        const int nl = n < p.N ? n : 0;                        // N % 32 == 0: a lane's 4 columns are in or out together
        a_tile = (const char*)p.x + (size_t)m0 * p.K * 2;
        if constexpr (BITS == 3) b_lane_off = ((unsigned)nl >> 4) * (unsigned)p.chunks * 768u + (unsigned)half * 192u + ((unsigned)nl & 15u) * 12u;
        else if constexpr (BITS == 8) b_lane_off = ((unsigned)nl >> 4) * (unsigned)p.chunks * 1024u + (unsigned)half * 512u + ((unsigned)nl & 15u) * 16u;
        else b_lane_off = ((unsigned)nl >> 4) * (unsigned)p.chunks * 1024u + (unsigned)half * 256u + ((unsigned)nl & 15u) * 16u;    // strip, k-slot, column
        if constexpr (BITS == 4) {
            s_lane_off = (unsigned)nl * 2u;
            z_lane_off = ((unsigned)nl >> 3) * 4u;
            zsh = ((unsigned)nl & 7u) * 4u;
        } else {                                               // the strip's records; 16 scales, then 16 zero-point bytes
            s_lane_off = ((unsigned)nl >> 4) * (unsigned)p.groups * REC + ((unsigned)nl & 15u) * 2u;
            z_lane_off = ((unsigned)nl >> 4) * (unsigned)p.groups * REC + 32u + ((unsigned)nl & 15u) * (BITS == 8 ? 2u : 1u);
        }
        if constexpr (G32) { s_lane_off += (unsigned)half * srow_bytes; z_lane_off += (unsigned)half * zrow_step; }
        kt0 = 2 * (kp * p.upt + j0);
        kt1 = kt0 + 2 * len;
        kt_last = kt1 - 1;
        dma_a4(kt0, 0, 0);
        dma_a4(kt0, 0, 1);
        if constexpr (BITS == 8) { load_b8(kt0, b0, 0u); load_b8(kt0, b1, 1u); }
        else load_b(kt0, b0);
        load_c(kt0, c0);
    };
    open_segment();
    for (;;) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        wait_vmcnt<0>();
        __syncthreads();
        setup_dq(dq_cur, c0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (BITS == 8) bq_first[nt] = dq_cur.frag(b0.w[nt][0], b0.w[nt][1], nt);
            else bq_first[nt] = frag_of(dq_cur, b0, nt, 0);
        }

        for (int kt = kt0; kt < kt1; kt += 2) {                 // a unit is a whole 128-deep chunk: no conditional second step
            if constexpr (BITS == 8) {
                Use8 ua{b0, b1, b2}, ub{b2, b0, b1};
                step(kt, std::integral_constant<int, 0>{}, ua, ua, c1);          // -> lo = b2, hi = b0 (b1 free)
                step(kt + 1, std::integral_constant<int, 1>{}, ub, ub, c0);      // -> lo = b1, hi = b2 (b0 free)
                const WRaw<8> t = b0;
                b0 = b1; b1 = b2; b2 = t;
            } else {
                step(kt, std::integral_constant<int, 0>{}, b0, b1, c1);
                step(kt + 1, std::integral_constant<int, 1>{}, b1, b0, c0);
            }
        }

        // the finished piece, then the next segment's first loads, then the piece's epilogue
        const int e_t = t, e_m0 = m0, e_mlo = bm * 128, e_bn = bn;
        const bool head = j0 == 0, complete = j0 + len == p.upt;
        u0 += len;
        const bool more = u0 < u1;
        if (more) open_segment();
        // everything the epilogue addresses is derived from this copy of the lane id, which the compiler cannot see through: its ~100 address
        // computations are loop invariants it would otherwise hoist over the K loop and spill (all 512 registers are taken there)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        int nb = 0;                                             // published pieces of this tile: workgroups Lb + 1 .. Lb + nb
        if (head && !complete) {
            const int te = (e_t + 1) * p.upt;
            int b = Lb + 1;
            while (range_start(b + 1) < te) ++b;
            nb = b - Lb;
            if (tid < nb) {                                     // bounded waits (the publishers have nothing in front of their publish)
                unsigned* const f = p.flags + Lb + 1 + tid;
                for (unsigned spins = 0;; ++spins) {
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    if (spins > p.max_spins) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // a later piece of a tile somebody else finishes is published without waiting for anybody; the head piece's holder finishes the tile
        finish(!head, e_m0, e_mlo, e_bn, lane_e, nb);
        if (!more) break;
    }
}

}  // namespace wide
}  // namespace gptq
