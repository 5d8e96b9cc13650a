// gemm_panel_kernel.cuh -- short prompts and large batches (129 ... ~767 rows), round 6: whole-K PANELS of the decode copy on the 32x32x16 matrix core.
// See gemm_panel.hip for the why; this header holds the kernel body so that the bit widths / group modes can live in separate translation units.
//
// Workgroup = one tile of R = 32 MT rows x 32 NT columns x the WHOLE K: nothing is exchanged between workgroups (no flags, no published pieces, no fix-up),
// so the launch has no fixed cost besides its own prologue and epilogue.  Its KP waves are K PARTS: wave w accumulates the 64-deep steps [w spw, (w + 1) spw)
// of the tile in its own registers from its own x buffers (no barrier in the K loop) and the parts meet ONCE, through LDS, in a fixed order
// (bit-reproducible).  Per 64-deep step a wave
//   * DMAs its R rows x 128 bytes of x into one of its two private LDS buffers (global_load_lds_dwordx4, 8 rows per instruction, 16-byte pieces XOR-swizzled
//     on the GLOBAL side so that the fragment reads below hit all 64 banks: the swizzle of gemm_wide_sk_kernel.cuh),
//   * loads, for each of its NT column blocks, the lane's 16 bytes of the decode copy -- lane (c = lane & 31, h = lane >> 5) of block nt owns column
//     n0 + 32 nt + c and k-slot 2 (kt & 1) + h of chunk kt / 2: the 4 words = the 4 MFMA steps of the 64-deep step, k = 32 h + 8 ks + 0..7 -- and the
//     column's scale and zero-point (as used) from the copy's constant records: a wave load instruction touches 4 runs of 256 contiguous bytes,
//   * dequantises (the exact magic-number form: w - z in packed fp16, times the scale with ONE rounding: bit-exact W) and runs MT x NT x 4
//     v_mfma_f32_32x32x16: A = x fragment (row 32 mt + c, the same k as above) read from LDS, B = the dequantised column.
// Loads are inline asm with hand-counted vmcnt (gemm_rows_kernel.cuh explains why): queue of a wave, oldest first, at the top of step kt =
// [W(kt), X(kt), W(kt + 1)] -> s_waitcnt vmcnt(loads of one W) leaves exactly W(kt + 1) in flight; inside the step X(kt + 1) is issued FIRST (under MFMA steps
// 0 and 1, half each) and W(kt + 2) behind it (DW = 3 register sets: at step 2; DW = 2: at the end of the step, into the set it just consumed), so the weights
// (HBM / far L2) get almost two steps of latency and x (L2) one.  Steps past the wave's range are clamped to its last step (redundant loads, no branches).
#pragma once
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"
#include "gemm_rows_kernel.cuh"      // rowsk::Deq1<T>: one column's constants, one word (8 weights) at a time

namespace gptq {
namespace panel {

struct PanelParams {
    const unsigned* qweight;      // the layer's decode copy (qweight_tiled)
    const char* qconst;           // its constant records (qconst_tiled): [strip][group][48 bytes]
    const void* bias;
    const void* x;                // [M][K] (act-order layers: permuted in natural order of the re-sequenced rows by the pre-pass)
    void* out;
    int M, K, N;
    int nbm, nbn;                 // row tiles, column tiles
    int chunks;                   // chunks of the copy per strip (128-deep; 8 bits: 64-deep)
    int groups;
    int gsh;                      // group of the 64-deep step kt = min(kt >> gsh, groups - 1)   (32-wide groups: 2 kt + lane half)
    int steps, spw;               // K / 64; steps per wave
};

template <int KP> constexpr int units_per_batch() { return 128 / KP; }      // 1 KiB per (unit, wave): 128 KiB of LDS per batch of the cross-wave sum

template <typename T, int BITS, int MT, int NT, int KP, bool G32>
__device__ __forceinline__ void panel_body(const PanelParams& p) {
    constexpr int R = 32 * MT, XB = R * 128, NX = 4 * MT;
    // the decode copy per packing (gptq_mi355x.h): 4 bits: 128-deep chunks of 1024 bytes, a lane's k-slot = 16 bytes; 3 bits: 768-byte chunks, 12 bytes; 8 bits: 64-deep
    // chunks of 1024 bytes, a k-slot = 16 k = 16 bytes -- the lane's 32 k of a step are k-slots 2 half, 2 half + 1 of chunk kt: two 16-byte loads
    constexpr int NH = BITS == 8 ? 2 : 1;                      // 16-byte (12-byte) loads per column block and step
    constexpr unsigned STEP_B = BITS == 3 ? 384u : (BITS == 8 ? 1024u : 512u), SLOT_B = BITS == 3 ? 192u : (BITS == 8 ? 512u : 256u), COL_B = BITS == 3 ? 12u : 16u;
    constexpr unsigned STRIP_CH = BITS == 3 ? 768u : 1024u, REC = BITS == 8 ? 64u : 48u;
    constexpr int NWL = NT * NH;                               // weight loads of a step; + 2 NT where constants are loaded
    constexpr int DW = 3;                                      // register sets of packed weights: W(kt + 2) is issued inside step kt
    constexpr int UNITS = 4 * MT * NT;                         // float4s per lane of the accumulator tile
    constexpr int UB = units_per_batch<KP>() < UNITS ? units_per_batch<KP>() : UNITS;
    static_assert(UB % KP == 0 || UB == UNITS, "every wave sums the same number of units per batch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int bm = Lb % p.nbm, bn = Lb / p.nbm;               // consecutive workgroups (one XCD): the same columns (weights), the next rows
    // (bands of 4 row tiles -- an XCD = 4 x 8 instead of 8 x 4 tiles, 20 % less L2 fill: measured within 1 %, tools/session_r06_panel2.sh)
    const int m0 = max(0, min(bm * R, p.M - R)), m_lo = bm * R;   // the last row tile is shifted up to end at row M and stores only its own rows
    // Fewer than R rows in all (33 .. 63 rows on wide layers: one partial panel): the x DMAs past the last row re-read it -- DMA i fetches rows 8 min(i, imax) +
    // min(r8, rlast) -- so nothing outside x is touched; the rows of the tile past M hold a copy of row M - 1 and are not stored.  A full tile: imax = rlast = 7, the
    // same addresses as before.
    // (tiles of up to three column blocks: the four-block forms live in all 256 registers and take 64+ rows only -- plan_panel)
#ifdef GPTQ_PANEL_NO_PART                                      // lab: the kernel as it was before partial panels (same-session A/B of the full-tile path)
    constexpr bool PART = false;
#else
    constexpr bool PART = NT <= 3;
#endif
    const int rows_here = PART ? min(R, p.M) : R;
    const int imax = __builtin_amdgcn_readfirstlane((rows_here - 1) >> 3), rlast = (rows_here - 1) & 7;
    const int n0 = bn * 32 * NT;

    const char* wbase[NT];
    const char* cbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int nb = n0 + 32 * nt;
        if (nb >= p.N) nb = n0;                                // a column block past N (N % 32 == 0: in or out as a whole) re-reads block 0 and is not stored
        wbase[nt] = (const char*)p.qweight + (size_t)(nb >> 4) * p.chunks * STRIP_CH;
        cbase[nt] = p.qconst + (size_t)(nb >> 4) * p.groups * REC;
    }
    const unsigned wlane = (unsigned)(l31 >> 4) * (unsigned)p.chunks * STRIP_CH + (unsigned)half * SLOT_B + (unsigned)(l31 & 15) * COL_B;
    const unsigned clane = (unsigned)(l31 >> 4) * (unsigned)p.groups * REC + (G32 ? (unsigned)half * REC : 0u);
    const unsigned slane = clane + (unsigned)(l31 & 15) * 2u, zlane = clane + 32u + (unsigned)(l31 & 15) * (BITS == 8 ? 2u : 1u);

    char* const xbuf = smem + (size_t)wave * 2 * XB;
    const unsigned xbuf_lds = lds_addr_of(xbuf);
    // x DMA i (8 rows x 128 bytes): lane (r8 = lane >> 3, kc = lane & 7) lands at LDS row 8 i + r8, slot kc, and fetches piece kc ^ (((8 i + r8) >> 1) & 7)
    unsigned xoff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const unsigned r8 = (unsigned)lane >> 3, kc = (unsigned)lane & 7u;
        xoff[par] = r8 * (unsigned)p.K * 2u + ((kc ^ (4u * par + (r8 >> 1))) * 16u);
    }
    unsigned xoff_last[2];
    if constexpr (PART) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const unsigned r8 = (unsigned)lane >> 3, kc = (unsigned)lane & 7u;
            xoff_last[par] = min(r8, (unsigned)rlast) * (unsigned)p.K * 2u + ((kc ^ (4u * par + (r8 >> 1))) * 16u);      // (the swizzle follows the LDS row, the address the clamped source row)
        }
    }
    const char* const xrow0 = (const char*)p.x + (size_t)m0 * p.K * 2;
    const size_t x8 = (size_t)8 * p.K * 2;
    // A fragment of row block mt, MFMA step ks: piece 4 half + ks of row 32 mt + l31, stored at slot piece ^ ((row >> 1) & 7)
    unsigned aoff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aoff[ks] = (unsigned)(l31 * 128 + (((half * 4 + ks) ^ ((l31 >> 1) & 7)) * 16));

    using WV = std::conditional_t<BITS == 3, wide::u32x3, u32x4>;      // a lane's words of one load (written by ONE instruction)
    struct Buf { WV w[NT][NH]; unsigned cs[NT], cz[NT]; };
    Buf q[DW];
    // Every constant register starts as ITS OWN opaque definition: initialised from one shared zero, the compiler kept scale and zero-point of a set in ONE register
    // up to the first load and split them with a v_mov BEHIND that load's asm -- a copy of a register whose load had not landed (8-bit g64 form: outputs that
    // differed from run to run).  tools/isa_inflight_lint.py checks the built kernels for any read of a load's destination in front of the next s_waitcnt.
#pragma unroll
    for (int j = 0; j < DW; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            asm volatile("v_mov_b32 %0, 0" : "=v"(q[j].cs[nt]));
            asm volatile("v_mov_b32 %0, 0" : "=v"(q[j].cz[nt]));
        }
    const int k0 = wave * p.spw, k1 = min(k0 + p.spw, p.steps);
    // constants only where a group begins (and at the wave's first step): the two sub-dword loads per column cost the memory pipe as much as the 16-byte weight
    // load beside them (profiles/r06_panel_ablate.log).  "+v": a step without constants leaves the registers as they are -- no copy can appear at the join
    const int gmask = (1 << p.gsh) - 1;
    auto has_c = [&](int kt) -> bool { return G32 || kt == k0 || (kt & gmask) == 0; };
    auto issue_w = [&](int kt, Buf& B) __attribute__((always_inline)) {
#if defined(GPTQ_PANEL_ABL) && (GPTQ_PANEL_ABL & 4)      // lab: 4 = no weight / constant loads
        if (kt != 0x7fffffff) return;
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const char* wsrc = wbase[nt] + (size_t)kt * STEP_B;      // 3 / 4 bits: chunk kt / 2, k-slots 2 (kt & 1) + half; 8 bits: chunk kt, k-slots 2 half, + 1
            if constexpr (BITS == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(B.w[nt][0]) : "v"(wlane), "s"(wsrc) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(B.w[nt][0]) : "v"(wlane), "s"(wsrc) : "memory");
            if constexpr (BITS == 8) asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" : "=v"(B.w[nt][1]) : "v"(wlane), "s"(wsrc) : "memory");
        }
        if (has_c(kt)) {
            const int g = G32 ? 2 * kt : min(kt >> p.gsh, p.groups - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const char* csrc = cbase[nt] + (size_t)g * REC;
                asm volatile("global_load_ushort %0, %1, %2" : "+v"(B.cs[nt]) : "v"(slane), "s"(csrc) : "memory");
                if constexpr (BITS == 8) asm volatile("global_load_ushort %0, %1, %2" : "+v"(B.cz[nt]) : "v"(zlane), "s"(csrc) : "memory");
                else asm volatile("global_load_ubyte %0, %1, %2" : "+v"(B.cz[nt]) : "v"(zlane), "s"(csrc) : "memory");
            }
        }
    };
    auto issue_x = [&](int kt, int buf, int i0, int i1) __attribute__((always_inline)) {      // DMAs [i0, i1) of step kt's rows into buffer buf
#if defined(GPTQ_PANEL_ABL) && (GPTQ_PANEL_ABL & 2)      // lab: 2 = no x DMAs
        if (kt != 0x7fffffff) return;
#endif
        const char* xsrc = xrow0 + (size_t)kt * 128;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(xbuf_lds + (unsigned)(buf * XB));
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            if constexpr (PART) {
                const unsigned xo = i >= imax ? xoff_last[i & 1] : xoff[i & 1];
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (unsigned)(i * 1024)), "v"(xo), "s"(xsrc + (size_t)min(i, imax) * x8) : "memory");
            } else {
                const unsigned xo = xoff[i & 1];
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (unsigned)(i * 1024)), "v"(xo), "s"(xsrc + (size_t)i * x8) : "memory");
            }
        }
    };
    // the registers pass through a statement behind the wait so that no use of them is scheduled in front of it
    auto claim = [&](Buf& B) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int h = 0; h < NH; ++h) asm volatile("" : "+v"(B.w[nt][h])::"memory");
            asm volatile("" : "+v"(B.cs[nt]), "+v"(B.cz[nt])::"memory");
        }
    };

    typename rowsk::DeqSel<T, BITS>::type dq[NT];              // the current group's constants (set up where a group begins)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    if (k0 < k1) {
        const int kl = k1 - 1;
        issue_w(k0, q[0]);
        issue_x(k0, 0, 0, NX);
        issue_w(min(k0 + 1, kl), q[1]);
        for (int kb = k0; kb < k1; kb += DW) {
#pragma unroll
            for (int j = 0; j < DW; ++j) {
                const int kt = kb + j;
                if (kt >= k1) break;
                const int buf = (kt - k0) & 1;
                const int ktx = min(kt + 1, kl), ktw = min(kt + 2, kl);
#if !(defined(GPTQ_PANEL_ABL) && (GPTQ_PANEL_ABL & 1))      // lab (tools/ab_unit.sh ... -DGPTQ_PANEL_ABL=n; wrong results by construction): 1 = nobody waits for the step's loads
                if (has_c(ktx)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWL + 2 * NT) : "memory");      // W(kt + 1) stays in flight: with or without constants
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWL) : "memory");
#endif
                claim(q[j]);
                if (has_c(kt)) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) dq[nt].setup(q[j].cs[nt], q[j].cz[nt]);
                }
                const char* xb = xbuf + buf * XB;
                // the 8 weights k = 32 half + 8 ks + 0..7 of column block nt, dequantised (bit-exact W)
                auto frag_of = [&](int nt, int ks) __attribute__((always_inline)) -> u32x4 {
                    if constexpr (BITS == 4) return dq[nt].frag(q[j].w[nt][0][ks]);
                    else if constexpr (BITS == 3) return dq[nt].frag(q[j].w[nt][0], ks);
                    else return dq[nt].frag(q[j].w[nt][ks >> 1][2 * (ks & 1)], q[j].w[nt][ks >> 1][2 * (ks & 1) + 1]);
                };
                u32x4 a[2][MT], bq[2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[0][mt] = *(const u32x4*)(xb + mt * 4096 + aoff[0]);
                bq[0] = frag_of(0, 0);
                // software pipeline over the 4 NT (MFMA step, column block) pairs: the NEXT pair's B fragment (13 VALU) is dequantised between the MT MFMAs of
                // this pair -- independent work for the 32 cycles each MFMA holds the matrix pipe (back to back, the second MFMA of a pair stalls the wave
                // for the first one's passes and the dequant then runs with the pipe idle)
#pragma unroll
                for (int i = 0; i < 4 * NT; ++i) {
                    const int ks = i / NT, nt = i % NT;
                    if (nt == 0) {
                        __builtin_amdgcn_sched_barrier(0);     // (the scheduler otherwise strings the MFMAs of one accumulator across the steps: dependent chains)
                        if (ks + 1 < 4) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) a[(ks + 1) & 1][mt] = *(const u32x4*)(xb + mt * 4096 + aoff[ks + 1]);
                        }
#if defined(GPTQ_PANEL_LAB_XISSUE) && GPTQ_PANEL_LAB_XISSUE == 1      // lab: all of the next step's x DMAs under MFMA step 0, the weights under step 1
                        if (ks == 0) issue_x(ktx, buf ^ 1, 0, NX);
                        if (DW == 3 && ks == 1) issue_w(ktw, q[(j + 2) % DW]);
#elif defined(GPTQ_PANEL_LAB_XISSUE) && GPTQ_PANEL_LAB_XISSUE == 2    // lab: a third of the DMAs per MFMA step 0..2, the weights under step 3
                        if (ks == 0) issue_x(ktx, buf ^ 1, 0, 3);
                        if (ks == 1) issue_x(ktx, buf ^ 1, 3, 6);
                        if (ks == 2) issue_x(ktx, buf ^ 1, 6, NX);
                        if (DW == 3 && ks == 3) issue_w(ktw, q[(j + 2) % DW]);
#else
                        if (ks == 0) issue_x(ktx, buf ^ 1, 0, NX / 2);
                        if (ks == 1) issue_x(ktx, buf ^ 1, NX / 2, NX);
                        if (DW == 3 && ks == 2) issue_w(ktw, q[(j + 2) % DW]);
#endif
#if defined(GPTQ_PANEL_LAB_PRIO)                                    // lab: the wave inside its MFMA pairs outranks its SIMD partner
                        if (ks == 0) __builtin_amdgcn_s_setprio(1);
#endif
                    }
#if defined(GPTQ_PANEL_ABL) && (GPTQ_PANEL_ABL & 8)      // lab: 8 = no dequant math
                    if (i + 1 < 4 * NT) { const unsigned qq = q[j].w[(i + 1) % NT][0][((i + 1) / NT) % 3]; bq[(i + 1) & 1] = u32x4{qq, qq ^ 0x11111111u, qq ^ 0x22222222u, qq ^ q[j].cs[(i + 1) % NT]}; }
#else
                    if (i + 1 < 4 * NT) bq[(i + 1) & 1] = frag_of((i + 1) % NT, (i + 1) / NT);
#endif
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = wide::Mma<T>::run(a[ks & 1][mt], bq[i & 1], acc[mt][nt]);
                    if (i + 1 < 4 * NT) {                     // MFMA, its share of the next fragment's 13 VALU, MFMA, ...
                        if constexpr (MT == 2) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                }
#if defined(GPTQ_PANEL_LAB_PRIO)
                __builtin_amdgcn_s_setprio(0);
#endif
                if (DW == 2) issue_w(ktw, q[j]);
            }
        }
        // The clamped loads of the last steps are still in flight and nothing below reads their registers: to the compiler those registers are dead from
        // here on and it would hand them to the epilogue (the accumulators copied out of the AGPRs landed in them and were overwritten by the late loads:
        // the first GPU run of this kernel).  Every register set passes through a statement BEHIND the wait, so it stays allocated until the loads are in.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < DW; ++j) claim(q[j]);
    }

    // ---- the K parts meet in LDS (fixed order), bias, store ------------------------------------------------------------------------------------
    // unit u = (mt, nt, rq): the float4 acc[mt][nt][4 rq .. 4 rq + 3] of every lane = rows 32 mt + 8 rq + 4 half + 0..3 of column n0 + 32 nt + l31 (C/D layout
    // of the 32x32 MFMA).  A batch of UB units: every wave writes its UB float4s ([unit][wave][lane]: 1 KiB per unit and wave), then wave w sums units
    // w, w + KP, ... over all KP waves in wave order -- it reads its own contribution from LDS too, so no accumulator register is ever indexed by the wave id.
    __syncthreads();                                           // (every wave's x DMAs have landed -- the wait above: the buffers are about to be overwritten)
    float* const red = (float*)smem;
#pragma unroll
    for (int b0 = 0; b0 < UNITS; b0 += UB) {
        const int b1 = b0 + UB < UNITS ? b0 + UB : UNITS;
        if (b0) __syncthreads();
#pragma unroll
        for (int u = b0; u < b1; ++u) {
            const int mt = u / (4 * NT), nt = (u / 4) % NT, rq = u % 4;
            const f32x16& a = acc[mt][nt];
            const f32x4 v = {a[rq * 4], a[rq * 4 + 1], a[rq * 4 + 2], a[rq * 4 + 3]};
            *(f32x4*)(red + ((size_t)((u - b0) * KP + wave) * 64 + lane) * 4) = v;
        }
        __syncthreads();
        for (int u = b0 + wave; u < b1; u += KP) {
            f32x4 v = *(const f32x4*)(red + ((size_t)((u - b0) * KP) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < KP; ++w) v += *(const f32x4*)(red + ((size_t)((u - b0) * KP + w) * 64 + lane) * 4);
            const int mt = u / (4 * NT), nt = (u / 4) % NT, rq = u % 4;
            const int n = n0 + 32 * nt + l31;
            if (n >= p.N) continue;
            const float bv = p.bias ? DType<T>::to_f32(((const T*)p.bias)[n]) : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 32 * mt + 8 * rq + 4 * half + i;
                if (m >= m_lo && m < p.M) ((T*)p.out)[(size_t)m * p.N + n] = DType<T>::from_f32(v[i] + bv);
            }
        }
    }
}

template <typename T, int BITS, int NT, bool G32>
__global__ void __launch_bounds__(512, 1) gemm_panel_kernel(PanelParams p) {
    panel_body<T, BITS, 2, NT, 8, G32>(p);
}

}  // namespace panel
}  // namespace gptq
