// gemm_wide_sk.hip -- prefill, round 5: the 128 x 128-per-wave tile of gemm_wide.hip for launches that do NOT divide into whole rounds of 128 x 512 tiles
// (BASELINE config 3: M = 2048 on the Llama-7B shapes).  One PERSISTENT workgroup per CU walks a contiguous range of (tile, K-chunk) units -- a stream-K
// partition, the schedule Marlin uses for the same reason (reference: autogptq_extension/marlin/marlin_cuda_kernel.cu:234-300 stripes, :580-660 the
// cross-block reduction) -- so every CU gets the same number of 128-deep K-chunks whatever the tile count.
//
// Tile = 128 rows x 256 columns; the four waves of a workgroup are 2 column halves x 2 K PARTS: wave (cw, kp) owns the 128 x 128 tile of columns
// [128 cw, 128 cw + 128) and accumulates the K-chunks of part kp (the tile's K range cut in two); the two parts are summed through LDS when the tile is
// complete.  That halves the tile count a launch needs to fill the chip (M = 2048 on 4096 -> 4096: 256 tiles = one per CU, no fix-up at all) and
// costs 8 instead of 4 x-tile DMAs per thread and K-step (each K part stages its own 128 x 64 x tile).
// A unit = one K-chunk pair: chunk j of part 0 and chunk upt + j of part 1 (upt = K / 256 units per tile).  Workgroup b runs units
// [b U / G, (b + 1) U / G) in ascending (tile, j) order.  A tile cut between workgroups is finished by the workgroup that holds its HEAD piece (j = 0 ...) --
// which it reaches LAST in its range -- while every other piece is the FIRST thing its workgroup runs: that workgroup publishes its fp32 accumulators
// (write-through stores, 64 KiB per wave) and raises a flag without ever waiting for anybody, so a finisher only waits for workgroups that have nothing
// in front of their publish (no residency assumption beyond "every workgroup is eventually scheduled"; the grid is one workgroup per CU anyway).  Sums
// are formed in a fixed order (own part + partner part through LDS, then the published pieces in range order, part 0 before part 1): bit-reproducible.
// The last row tile is SHIFTED UP to end at row M (m0 = M - 128) instead of clamping rows: no x address of the DMA ever needs a per-lane clamp (M >= 128);
// the overlapped rows are computed twice and stored once, by the tile they belong to.
// Weights: the layer's decode copy (gptq_prepack_decode), raw x by LDS DMA, the exact magic-number dequant -- the loop body is gemm_wide_kernel<T, true,
// true, G128>'s (gemm_wide.hip), minus its ablation switches.
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"
#include "launch.h"

namespace gptq {
namespace wide {

struct WskParams {
    const unsigned* qweight;      // the layer's decode copy (qweight_tiled)
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
    float* slots;                 // [workgroup][wave][64 quads][64 lanes] float4: the accumulators of a published piece
    unsigned* err;                // sticky error word: a bounded wait gave up
};

constexpr int WSK_XT_BYTES = 128 * 128;                       // one x tile: 128 rows x 64 k x 2 bytes
constexpr int WSK_LDS_BYTES = 4 * 32768;                      // the exchange area (4 waves x 32 KiB) over the four x tiles (2 buffers x 2 K parts x 16 KiB)
constexpr size_t WSK_SLOT_FLOATS = (size_t)4 * 64 * 64 * 4;   // one workgroup's published piece: 256 KiB

template <typename T, bool G128>
__global__ void __launch_bounds__(256, 1) gemm_wide_sk_kernel(WskParams p) {
    constexpr int KS = 4, MT = 4, NT = 4, STRIDE = 128;
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
    const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qweight, 0, (int)((size_t)(p.N / 16) * p.chunks * 1024), 0x00020000);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, 0, (int)((size_t)p.groups * p.N * 2), 0x00020000);
    const int zrow_bytes = p.N / 8 * 4;
    const auto rsrc_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.qzeros, 0, p.groups * zrow_bytes, 0x00020000);
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;
    const int a_lane_off = l31 * STRIDE;
    const int a_swz = (l31 >> 1) & 7;

    // per-tile state (set at the top of every segment)
    const char* a_tile = nullptr;                              // x + m0 * K
    unsigned b_lane_off = 0, s_lane_off = 0, z_lane_off = 0, zsh = 0;
    int kt_last = 0;

    auto dma_a4 = [&](int kt, int buf, int q) {                // DMAs 4 q .. 4 q + 3 of the wave's eight
        const char* sb = a_tile + (size_t)kt * 128 + (size_t)(cw * 8 + q * 4) * a_grp_bytes;
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
    auto load_b = [&](int kt, u32x4 (&b)[KS]) {
        const unsigned so = (unsigned)(kt >> 1) * 1024u + (unsigned)(kt & 1) * 512u;
#pragma unroll
        for (int col = 0; col < NT; ++col) b[col] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, b_lane_off + col * 16u, so, 0);
    };
    auto load_c = [&](int kt, CRaw& c) {
        const int g = (int)(((unsigned long long)(unsigned)kt * p.kpg_inv) >> 32);
        c.s = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_s, s_lane_off, (unsigned)g * (unsigned)p.N * 2u, 0));
        c.z = __builtin_amdgcn_raw_buffer_load_b32(rsrc_z, z_lane_off, (unsigned)(g * zrow_bytes), 0);
    };

    f32x16 acc[MT][NT];
    u32x4 b0[KS], b1[KS];
    CRaw c0, c1;
    Deq4<T> dq_cur;
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
    auto step = [&](int kt, auto bufc, const u32x4 (&b_use)[KS], u32x4 (&b_fill)[KS], CRaw& c_fill) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool NEWG = !(G128 && BUF == 0);             // does step kt + 1 open a new group?  (G128: only behind the odd step of a body)
        const int ktn = min(kt + 1, kt_last);                  // the segment's last step re-loads itself (no branch in the pipeline)
        // claim this step's weight words before anything new is issued: the compiler's exact wait lands here (gemm.hip)
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) asm volatile("" ::"v"(b_use[ks][0]), "v"(b_use[ks][1]), "v"(b_use[ks][2]), "v"(b_use[ks][3]));
        __builtin_amdgcn_sched_barrier(0);
        const char* abase = smem + (size_t)(BUF * 2 + kp) * WSK_XT_BYTES + a_lane_off;
        u32x4 a[2][MT], bq[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[0][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (((half * 4 + 0) ^ a_swz) * 16));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[0][nt] = bq_first[nt];
        __builtin_amdgcn_sched_barrier(0);
        Deq4<T> dq_nx;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[(ks + 1) & 1][mt] = *(const u32x4*)(abase + mt * 32 * STRIDE + (((half * 4 + ks + 1) ^ a_swz) * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[(ks + 1) & 1][nt] = dq_cur.frag(b_use[nt][ks + 1], nt);
            } else {
                if constexpr (NEWG) dq_nx.setup(c_fill, zsh, zmask);
                else dq_nx = dq_cur;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq_first[nt] = dq_nx.frag(b_fill[nt][0], nt);
            }
            if (ks == 0) {                                     // next step's weights + constants: under MFMA group 0
                load_b(ktn, b_fill);
                if constexpr (NEWG) load_c(ktn, c_fill);
            }
            if (ks == 1) dma_a4(ktn, BUF ^ 1, 0);              // next step's x tile: under groups 1 and 2
            if (ks == 2) dma_a4(ktn, BUF ^ 1, 1);
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
        wait_vmcnt<0>();                                       // the next tile must have landed before anybody passes the barrier (the DMAs are the step's newest VMEM operations)
        __syncthreads();
    };

    // exchange + output of a finished tile: this wave keeps row tiles OWN .. OWN + 1 and hands the other two to its partner (same columns, other K part)
    auto finish = [&](auto ownc, int m0, int m_lo, int bn, int lane_e, int nb) {
        constexpr int OWN = decltype(ownc)::value, OTHER = 2 - OWN;
        const int half_e = lane_e >> 5;
        const int n = bn * 256 + cw * 128 + 4 * (lane_e & 31);
        const bool col_ok = n < p.N;
        char* const ex_mine = smem + (size_t)wave * 32768;
        const char* const ex_partner = smem + (size_t)(wave ^ 2) * 32768;
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x16& a = acc[OTHER + mtl][nt];
                    const f32x4 v = {a[rq * 4], a[rq * 4 + 1], a[rq * 4 + 2], a[rq * 4 + 3]};
                    *(f32x4*)(ex_mine + (size_t)((((mtl * NT + nt) * 4 + rq) * 64 + lane_e) * 16)) = v;
                }
        __syncthreads();                                       // ... and thread 0's flag waits are behind everybody
        float bias[NT] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && col_ok) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bias[nt] = DType<T>::to_f32(((const T*)p.bias)[n + nt]);
        }
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl) {
            const int mt = OWN + mtl;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x16& a = acc[OWN + mtl][nt];
                    const f32x4 mine = {a[rq * 4], a[rq * 4 + 1], a[rq * 4 + 2], a[rq * 4 + 3]};
                    const f32x4 theirs = *(const f32x4*)(ex_partner + (size_t)((((mtl * NT + nt) * 4 + rq) * 64 + lane_e) * 16));
                    v[nt] = (OWN == 0) ? (mine + theirs) : (theirs + mine);      // part 0 + part 1 on both sides (fp32 addition commutes: written for the reader)
                }
                for (int s = 0; s < nb; ++s) {                 // published pieces, in range order, part 0 before part 1
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const float* slot = p.slots + ((size_t)(Lb + 1 + s) * 4 + (size_t)(kk * 2 + cw)) * (64 * 64 * 4);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {      // agent-scope loads: they bypass this XCD's non-coherent L2 lines
                            const unsigned long long* src = (const unsigned long long*)(slot + (size_t)((((mt * NT + nt) * 4 + rq) * 64 + lane_e) * 4));
                            const unsigned long long w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v[nt][0] += __builtin_bit_cast(float, (unsigned)w0);
                            v[nt][1] += __builtin_bit_cast(float, (unsigned)(w0 >> 32));
                            v[nt][2] += __builtin_bit_cast(float, (unsigned)w1);
                            v[nt][3] += __builtin_bit_cast(float, (unsigned)(w1 >> 32));
                        }
                    }
                }
                if (col_ok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {              // C/D layout of the 32x32 MFMA: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), r = 4 rq + i
                        const int m = m0 + mt * 32 + i + 8 * rq + 4 * half_e;
                        if (m < m_lo) continue;                // the shifted last row tile stores only its own rows (the rows above belong to the tile before it)
                        const unsigned lo = (unsigned)t_bits(DType<T>::from_f32(v[0][i] + bias[0])) | ((unsigned)t_bits(DType<T>::from_f32(v[1][i] + bias[1])) << 16);
                        const unsigned hi = (unsigned)t_bits(DType<T>::from_f32(v[2][i] + bias[2])) | ((unsigned)t_bits(DType<T>::from_f32(v[3][i] + bias[3])) << 16);
                        *(u32x2*)((unsigned short*)p.out + (size_t)m * p.N + n) = u32x2{lo, hi};
                    }
                }
            }
        }
    };

    for (;;) {                                                 // segments: the pieces of tiles inside [u0, u1)
        const int t = u0 / p.upt;
        const int j0 = u0 - t * p.upt;
        const int len = min(p.upt - j0, u1 - u0);
        const int bm = t / p.nbn, bn = t - bm * p.nbn;
        const int m0 = min(bm * 128, p.M - 128);
        const int n = bn * 256 + cw * 128 + 4 * l31;          // this lane's first column (it owns n .. n + 3)
        const bool col_ok = n < p.N;                           // N % 32 == 0: a lane's 4 columns are in or out together
        const int nl = col_ok ? n : 0;
        a_tile = (const char*)p.x + (size_t)m0 * p.K * 2;
        b_lane_off = ((unsigned)nl >> 4) * (unsigned)p.chunks * 1024u + (unsigned)half * 256u + ((unsigned)nl & 15u) * 16u;    // strip, k-slot, column
        s_lane_off = (unsigned)nl * 2u;
        z_lane_off = ((unsigned)nl >> 3) * 4u;
        zsh = ((unsigned)nl & 7u) * 4u;
        const int kt0 = 2 * (kp * p.upt + j0), kt1 = kt0 + 2 * len;
        kt_last = kt1 - 1;

#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        dma_a4(kt0, 0, 0);
        dma_a4(kt0, 0, 1);
        load_b(kt0, b0);
        load_c(kt0, c0);
        wait_vmcnt<0>();
        __syncthreads();
        dq_cur.setup(c0, zsh, zmask);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq_first[nt] = dq_cur.frag(b0[nt][0], nt);

        for (int kt = kt0; kt < kt1; kt += 2) {                 // a unit is a whole 128-deep chunk: no conditional second step
            step(kt, std::integral_constant<int, 0>{}, b0, b1, c1);
            step(kt + 1, std::integral_constant<int, 1>{}, b1, b0, c0);
        }

        // everything the epilogues address is derived from this copy of the lane id, which the compiler cannot see through: their ~130 address
        // computations are loop invariants it would otherwise hoist over the K loop and spill (all 512 registers are taken there)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const bool head = j0 == 0, complete = j0 + len == p.upt;
        if (!head) {
            // a later piece of a tile somebody else finishes: publish, drain the write-through stores, raise the flag -- no wait on anybody
            float* const slot = p.slots + ((size_t)Lb * 4 + wave) * (64 * 64 * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x16& a = acc[mt][nt];
                        const f32x4 v = {a[rq * 4], a[rq * 4 + 1], a[rq * 4 + 2], a[rq * 4 + 3]};
                        // s_nop inside the string: nothing is padded behind an asm statement, and the next instruction may overwrite the data registers
                        // (dead to the compiler) while the store still reads them (gemm.hip, tools/tail_diag.py)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slot + (size_t)((((mt * NT + nt) * 4 + rq) * 64 + lane_e) * 4)), "v"(v) : "memory");
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.flags + Lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int nb = 0;                                         // published pieces of this tile: workgroups Lb + 1 .. Lb + nb
            if (!complete) {
                const int te = (t + 1) * p.upt;
                int b = Lb + 1;
                while (range_start(b + 1) < te) ++b;
                nb = b - Lb;
                if (tid < nb) {                                 // bounded waits (the publishers have nothing in front of their publish)
                    unsigned* const f = p.flags + Lb + 1 + tid;
                    for (unsigned spins = 0;; ++spins) {
                        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                        if (spins > p.max_spins) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (kp == 0) finish(std::integral_constant<int, 0>{}, m0, bm * 128, bn, lane_e, nb);
            else finish(std::integral_constant<int, 2>{}, m0, bm * 128, bn, lane_e, nb);
        }
        u0 += len;
        if (u0 >= u1) break;
        __syncthreads();                                       // the exchange area is the next segment's x tiles
    }
}

}  // namespace wide

// ---- host side -------------------------------------------------------------------------------------------------------------------------------
bool wide_sk_ok(const gptq_layer_t& L, int M) {
    if (L.bits != 4 || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16)) return false;
    if (L.qweight_tiled == nullptr || L.tiled_cols != GPTQ_STRIP_COLS) return false;
    if (L.K % 256 || L.group_size % 64 || L.N % 32 || L.epilogue != GPTQ_EPI_NONE) return false;      // two K parts of whole 128-deep chunks
    return M >= 128;
}

// Provisional rule (round 5, before the sweep of tools/wide_sk_ab.py): prefill row counts.
bool wide_sk_pays(const gptq_layer_t& L, int M) { return wide_sk_ok(L, M) && M >= 1024; }

WideSkGeom wide_sk_geom(const gptq_layer_t& L, int M) {
    WideSkGeom g{};
    g.nbm = (M + 127) / 128;
    g.nbn = (L.N + 255) / 256;
    g.upt = L.K / 256;
    const long units = (long)g.nbm * g.nbn * g.upt;
    g.units_total = (int)units;
    g.lg_nwg = 8;                                             // one workgroup per CU
    while (g.lg_nwg > 0 && (1L << g.lg_nwg) > units) --g.lg_nwg;
    // does any range boundary fall inside a tile?  (then pieces are published: 256 KiB per workgroup behind the permuted x)
    bool cut = false;
    for (int b = 1; b < (1 << g.lg_nwg) && !cut; ++b) cut = (((long)b * units) >> g.lg_nwg) % g.upt != 0;
    g.slot_bytes = cut ? ((size_t)1 << g.lg_nwg) * wide::WSK_SLOT_FLOATS * sizeof(float) : 0;
    return g;
}

template <typename T, bool G128>
static hipError_t grant_wsk() {
    return hipFuncSetAttribute((const void*)wide::gemm_wide_sk_kernel<T, G128>, hipFuncAttributeMaxDynamicSharedMemorySize, wide::WSK_LDS_BYTES);
}
hipError_t init_gemm_wide_sk_device() {
    hipError_t e = grant_wsk<f16, true>();
    if (e == hipSuccess) e = grant_wsk<f16, false>();
    if (e == hipSuccess) e = grant_wsk<bf16, true>();
    if (e == hipSuccess) e = grant_wsk<bf16, false>();
    return e;
}

hipError_t launch_gemm_wide_sk(const gptq_layer_t& L, const void* x, void* out, int M, void* ws_header, void* slots, hipStream_t st) {
    if (!wide_sk_ok(L, M)) return hipErrorInvalidValue;
    const WideSkGeom g = wide_sk_geom(L, M);
    if (g.slot_bytes && (!ws_header || !slots)) return hipErrorInvalidValue;
    wide::WskParams p{};
    p.qweight = L.qweight_tiled; p.qzeros = L.qzeros; p.scales = L.scales; p.bias = L.bias; p.x = x; p.out = out;
    p.M = M; p.K = L.K; p.N = L.N; p.zero_mode = L.zero_mode;
    p.nbm = g.nbm; p.nbn = g.nbn; p.upt = g.upt; p.units_total = g.units_total; p.lg_nwg = g.lg_nwg;
    p.chunks = L.K / 128;
    p.groups = (L.K + L.group_size - 1) / L.group_size;
    const unsigned long long kpg = (unsigned long long)(L.group_size / 64);
    p.kpg_inv = ((1ull << 32) + kpg - 1) / kpg;
    p.max_spins = 1u << 22;
    p.flags = (unsigned*)ws_header;
    p.slots = (float*)slots;
    p.err = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2 : nullptr;
    const dim3 grid(1 << g.lg_nwg), block(256);
    const bool g128 = L.group_size % 128 == 0;
    if (L.dtype == GPTQ_F16) {
        if (g128) hipLaunchKernelGGL((wide::gemm_wide_sk_kernel<f16, true>), grid, block, wide::WSK_LDS_BYTES, st, p);
        else hipLaunchKernelGGL((wide::gemm_wide_sk_kernel<f16, false>), grid, block, wide::WSK_LDS_BYTES, st, p);
    } else {
        if (g128) hipLaunchKernelGGL((wide::gemm_wide_sk_kernel<bf16, true>), grid, block, wide::WSK_LDS_BYTES, st, p);
        else hipLaunchKernelGGL((wide::gemm_wide_sk_kernel<bf16, false>), grid, block, wide::WSK_LDS_BYTES, st, p);
    }
    return hipGetLastError();
}

}  // namespace gptq
