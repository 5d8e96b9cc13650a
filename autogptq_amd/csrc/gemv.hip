// gemv.hip -- memory-bound decode path: out[M,N] = x[M,K] @ dequant(qweight) for small M.
//
// Replaces (reference, AutoGPTQ v0.8.0.dev0): VecQuant{2,3,4,8}MatMulKernel* in
// autogptq_extension/cuda_256/autogptq_cuda_kernel_256.cu:281-1437, q4_matmul_kernel in
// exllama/cuda_func/q4_matmul.cu:33-143 and gemm_half_q_half_gptq_kernel in
// exllamav2/cuda/q_gemm_kernel_gptq.cuh:39-194.  Nothing here is derived from that code; the
// design is CDNA4-first:
//
//  * A workgroup owns a COLUMN STRIP of CT = 4*LN output columns over a K range; a lane owns 4
//    adjacent columns (one 128-bit nontemporal load per packed row: 16 B/lane) and the 64/LN "row
//    slots" of a wave walk consecutive packed rows, so one wave-load covers WR rows x (16*LN)
//    contiguous bytes.  Logical strip ids are remapped so neighbouring strips share an XCD (one L2).
//  * Default kernel (4-bit, fp16/bf16, power-of-two rows per group): gemv_q4_f16_mfma_kernel.  Nothing is
//    staged in LDS before the math: every lane issues, in this order, its (scales, zeros) pair, the 16 B of x
//    per packed row (L2 resident) and its U consecutive packed rows -- for the Llama-7B shapes the whole
//    matrix is in flight at once (one HBM round trip + a reduction).  w - z is formed EXACTLY in packed
//    fp16 ((q & 0x000f000f) | 0x64006400 is 1024 + w; adding -(1024 + z) is exact) and the k-reduction runs
//    on the matrix core: v_mfma_f32_4x4x4_16b_f16 is 16 independent 4x4x4 products, one per aligned 4-lane
//    group = one packed row x 16 columns, so up to 4 rows of x cost the same as one.  The scale multiplies
//    the fp32 group sums once per (lane, group).  No fp16 accumulation anywhere.
//  * Other packings (2/3/8-bit, fp16/bf16): gemv_mfma_generic_kernel, same structure; 3- and 8-bit fp16 fields are decoded two at a
//    time with the fp16 magic number (MagicF16: 16-bit windows of the 3-bit stream at multiples of 15 bits line up in both halves of a
//    register), everything else field by field.  Act-order layers of these packings: x permuted once by the column-permute pre-pass.
//    fp32 I/O, raw (non-uniform) act-order g_idx, odd group sizes: gemv_generic_kernel (fp32 FMA, x in LDS).
//    (Round 1's LDS-staged kernel and the v_dot2 register kernel -- comparison variants behind tuning.path = 2 / 4 -- were retired in
//    round 6: no default plan could reach them; 4-bit fp16 layers whose groups are not a power of two go to gemv_generic_kernel.)
//  * act-order layers read the group-sorted side copy of qweight; x is gathered through perm[] (whole row
//    staged in LDS once, rows pulled with ds_bpermute), which is the whole cost of act-order on this path.
//  * K reduction: row slots by DPP rotates / ds_bpermute, waves through LDS in fixed order (the kernel's
//    only barrier), then (only if ksplit > 1) a second pass over fp32 partials.  No atomics: bit-reproducible.
#include <type_traits>

#include "common.cuh"
#include "launch.h"
#include "gemv_shared.cuh"

namespace gptq {


// ---- matrix-core GEMV: 4-bit, fp16, sequential groups -------------------------------------------
// Every lane loads its U packed rows, the x they multiply and one (scales, zeros) pair straight into registers; the k-reduction runs on the matrix core:
// v_mfma_f32_4x4x4_16b_f16 is 16 independent 4x4x4 products, one per aligned group of 4 lanes -- and an
// aligned group of 4 lanes here is exactly one packed row x 16 columns.  Lane j of a group supplies B = 4
// consecutive-slot k of ITS OWN column, lane i supplies A = the same 4 k of x row i, and lane j receives
// D[0..3][j]: the partial sums of its own column for 4 rows of x.  So up to 4 rows of x cost the same as
// one, the 16 v_dot2 per packed word become 8 MFMAs (on a pipe that runs beside the VALU), and the scale is
// applied once per (lane, group) to the fp32 sums instead of once per weight:
//     out[m, n] = sum_g  s[g, n] * ( sum_{k in g} x[m, k] * (w[k, n] - z[g, n]) )      (w - z exact in fp16)
// Occupancy: a 16-wave workgroup needs <= 64 VGPRs for two of them to share a CU (8 waves per SIMD); that is what
// keeps the second "round" of workgroups of wide layers (N = 11008: 688 strips) from serialising behind the first.
// PAIR = fused SILU_MUL epilogue: the workgroup walks K twice, once for its strip of the gate half and once for the same
// strip of the up half (p.pair_off columns further), and writes silu(gate) * up -- one launch and no [M, N] round trip
// for the gate/up pair of a gated MLP.
// PERM = act-order layer: weights come from the group-sorted copy and the 8 x values of a packed row are gathered
// through perm[] (two 16-byte index loads + eight 2-byte gathers per row, all L2 resident) straight into slot order.
// T = bf16: gfx950 has no packed bf16 arithmetic.  (q & 0x000f000f) | 0x43004300 is the bf16 pair (128 + w_lo, 128 + w_hi);
// each half is widened to fp32 (a shift / a mask), -(128 + z) is added in fp32 (exact: small integers) and the upper halves
// of the two results -- bf16(w - z), exact -- are packed back with one v_perm: 4 VALU per pair instead of 1, invisible in a
// latency-bound kernel.  (A first version fed 128 + w to the matrix core and subtracted (128 + z) * sum_k x_k per group
// afterwards; algebraically the same, but the cancellation between the two large terms breaks the one property every other
// kernel here has -- a layer whose fields equal their zero-points gives exactly 0 -- and failed the reference's
// "range/range/range" value pattern (tests/test_hpu_linear.py:107) with scales ~ 10^4.)
template <int LN, int MT, int U, bool PAIR = false, bool PERM = false, typename T = f16>
__global__ void __launch_bounds__(1024, (std::is_same_v<T, f16> && !PAIR && !PERM && ((U == 1 && MT <= 4) || (U == 2 && MT <= 2))) ? 8 : 4) gemv_q4_f16_mfma_kernel(GemvParams p) {
    constexpr bool BF = std::is_same_v<T, bf16>;
    unsigned m_lo, m_hi, magic;                                   // see the dequant below
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;
    constexpr int WR = 64 / LN, CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int H = PAIR ? 2 : 1;
    const int NH = PAIR ? p.pair_off : p.N;            // output columns
    const int n0 = strip * CT + cl * 4;
    const bool col_ok = n0 < NH;
    const int nbase = col_ok ? n0 : 0;
    constexpr int RG = (MT == 8) ? 2 : 1;              // groups of 4 x rows handled per pass (MT = 8: two MFMA sets)
    constexpr int MTR = (MT == 8) ? 4 : MT;            // rows reduced / stored per group
    const int m0 = blockIdx.z * (4 * RG);
    const int ub = blockIdx.y * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    // A operand: lane i of each 4-lane group carries x row m0 + i (clamped: surplus rows are never stored)
    const T* xrow[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) xrow[rg] = (const T*)p.x + (size_t)min(m0 + rg * 4 + (lane & 3), p.M - 1) * p.K;
    const T* __restrict__ scales = (const T*)p.scales;
    const int zrow_words = p.N >> 3;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;
    const int gshift = p.gu_shift;

    float acc[H][RG][4][MTR];           // [half][row group][column][row of x]: only the rows that are stored
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int m = 0; m < MTR; ++m) acc[h][rg][c][m] = 0.f;

    const int rows_per_iter = W * WR * U;
    // act-order, M = 1: the whole x row (perm[] points anywhere in [0, K)) goes to LDS first -- coalesced 16-byte loads
    // issued before anything else, so they return first; the per-row gather through perm[] then runs on the LDS (a 2-byte
    // gather from global costs the texture path about one address per cycle: 11008 cycles for K = 11008, as long as the
    // whole weight stream)
    constexpr bool XLDS = PERM && MT == 1 && LN == 4;
    constexpr int NXST = XLDS ? 4 : 1;                     // staging loads per thread: up to 4 x 1024 x 8 = 32768 values
    unsigned short* xlds = (unsigned short*)(smem + (size_t)W * H * (RG * MTR) * CT * sizeof(float));
    u32x4 xst[NXST];
    if constexpr (XLDS) {
#pragma unroll
        for (int v = 0; v < NXST; ++v) {
            const int kk = (v * (int)blockDim.x + tid) * 8;
            xst[v] = *(const u32x4*)(xrow[0] + min(kk, p.K - 8));
        }
    }
    bool x_staged = false;
#pragma unroll
    for (int h = 0; h < H; ++h) {
    const int nload = nbase + h * p.pair_off;
    for (int base = ub; base < ue; base += rows_per_iter) {
        const int u0 = base + (wave * WR + rs) * U;
        // loads return in issue order: the small L2-resident ones (scales, zeros, x) go first so that the math
        // on row j can start as soon as weight row j lands, instead of behind the whole weight burst
        const int g = min(u0, ue - 1) >> gshift;
        const u32x2 sraw = *(const u32x2*)(scales + (size_t)g * p.N + nload);
        const unsigned zw = p.qzeros[(size_t)g * zrow_words + (nload >> 3)] >> ((nload & 7) * 4);
        u32x4 q[U], xr[RG][U];
        auto load_q = [&]() {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int ul = min(u0 + j, ue - 1);
                q[j] = __builtin_nontemporal_load((const u32x4*)(p.qweight + (size_t)ul * p.N + nload));
            }
        };
        // act-order: the x gathers depend on the perm[] loads, so the weight loads are issued in between (they must
        // not wait behind that dependency); otherwise x first (see above)
        if constexpr (!PERM) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int j = 0; j < U; ++j) xr[rg][j] = *(const u32x4*)(xrow[rg] + (size_t)min(u0 + j, ue - 1) * 8);
        } else if constexpr (MT == 1 && LN == 4) {
            // M = 1: the wave gathers its WR*U = 16*U consecutive packed rows cooperatively -- every lane fetches the
            // 2*U consecutive x values at flat offset lane*2U of the window through perm (U/2 index loads + 2U two-byte
            // gathers instead of 10 loads per row), then each row pulls its four dwords from their owner lanes with
            // ds_bpermute.  The weight loads are issued between the index loads and the gathers that depend on them.
            constexpr int VPL = 2 * U;                                    // x values gathered per lane (U dwords)
            const int wrow0 = base + wave * WR * U;                       // first packed row of this wave's window
            const int flat = min(wrow0 * 8 + lane * VPL, ue * 8 - VPL);   // clamped: tail rows are masked by `live`
            int pi[VPL];
            if constexpr (U == 1) {
                const u32x2 t = *(const u32x2*)(p.perm + flat);
                pi[0] = (int)t[0]; pi[1] = (int)t[1];
            } else {
#pragma unroll
                for (int v = 0; v < VPL / 4; ++v) {
                    const u32x4 t = *(const u32x4*)(p.perm + flat + 4 * v);
                    pi[4 * v] = (int)t[0]; pi[4 * v + 1] = (int)t[1]; pi[4 * v + 2] = (int)t[2]; pi[4 * v + 3] = (int)t[3];
                }
            }
            load_q();
            if (!x_staged) {                                              // first iteration only (uniform)
#pragma unroll
                for (int v = 0; v < NXST; ++v) {
                    const int kk = (v * (int)blockDim.x + tid) * 8;
                    if (kk < p.K) *(u32x4*)(xlds + kk) = xst[v];
                }
                for (int kk = (NXST * (int)blockDim.x + tid) * 8; kk < p.K; kk += (int)blockDim.x * 8)   // small workgroup (deep K split), long K
                    *(u32x4*)(xlds + kk) = *(const u32x4*)(xrow[0] + kk);
                __syncthreads();
                x_staged = true;
            }
            unsigned mine[U];
#pragma unroll
            for (int v = 0; v < U; ++v) mine[v] = (unsigned)xlds[pi[2 * v]] | ((unsigned)xlds[pi[2 * v + 1]] << 16);
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int r = rs * U + j;                                 // row inside the window (0 .. 16*U-1)
                u32x4 t;
#pragma unroll
                for (int d = 0; d < 4; ++d)                               // dword d of row r = x[2d], x[2d+1]: flat dword r*4+d
                    t[d] = (unsigned)__builtin_amdgcn_ds_bpermute(((r * 4 + d) / U) * 4, (int)mine[(j * 4 + d) % U]);
                // natural (x0,x1)(x2,x3)(x4,x5)(x6,x7) -> slot order
                xr[0][j] = u32x4{__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u),
                                 __builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
            }
        } else {
            u32x4 pidx[U][2];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int* pp = p.perm + (size_t)min(u0 + j, ue - 1) * 8;
                pidx[j][0] = *(const u32x4*)pp;
                pidx[j][1] = *(const u32x4*)(pp + 4);
            }
            load_q();
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const u32x4 p0 = pidx[j][0], p1 = pidx[j][1];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    const unsigned short* xs = (const unsigned short*)xrow[rg];
                    // already in slot order (k0,k4)(k1,k5)(k2,k6)(k3,k7)
                    xr[rg][j] = u32x4{(unsigned)xs[p0[0]] | ((unsigned)xs[p1[0]] << 16), (unsigned)xs[p0[1]] | ((unsigned)xs[p1[1]] << 16),
                                      (unsigned)xs[p0[2]] | ((unsigned)xs[p1[2]] << 16), (unsigned)xs[p0[3]] | ((unsigned)xs[p1[3]] << 16)};
                }
            }
        }
        if constexpr (!PERM) load_q();

        f16x2 c1[4], c2[4];
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned z = (((zw >> (4 * c)) & 15u) + 1u) & zmask;
            c1[c] = as_f16x2(z * 0x00010001u + 0xE400E400u);        // -(1024+z)
            c2[c] = c1[c] + k960;                                   // -(64+z)
        }
        f32x4 accg[RG][4];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int c = 0; c < 4; ++c) accg[rg][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
#pragma unroll
        for (int j = 0; j < U; ++j) {
            u32x4 qv = q[j];
            if (u0 + j >= ue) qv = u32x4{0u, 0u, 0u, 0u};       // tail rows: x is zeroed below, value irrelevant
            const bool live = (u0 + j < ue);
            // x slots (k0,k4,k1,k5) and (k2,k6,k3,k7): the order the magic-number unpack produces
            u32x2 xa01[RG], xa23[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const u32x4 t = xr[rg][j];
                u32x2 a01, a23;
                if constexpr (PERM) {
                    a01 = u32x2{t[0], t[1]};
                    a23 = u32x2{t[2], t[3]};
                } else {
                    a01 = u32x2{__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u)};
                    a23 = u32x2{__builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
                }
                if (!live) { a01 = u32x2{0u, 0u}; a23 = u32x2{0u, 0u}; }
                xa01[rg] = a01;
                xa23[rg] = a23;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned qw = qv[c], q8 = qw >> 8;
                u32x2 b01, b23;
                // w - z exactly in packed fp16.  The masks and the magic number are opaque to the compiler (defined by asm
                // above), so (q & mask) | magic is selected as ONE v_and_or_b32 (mask in an SGPR, magic in a VGPR) instead of
                // v_and_b32 + v_or_b32 with two literals: 4 VALU fewer per word.
                // (not for the two-pass PAIR kernel: there the opaque form makes hipcc hoist the second pass's unpack in front of
                //  its loads' use and the [gate|up] launch drops from 2625 to 2455 GB/s -- A/B/C in one session, bench.py)
                const unsigned ml = PAIR ? 0x000f000fu : m_lo, mh = PAIR ? 0x00f000f0u : m_hi, mg = PAIR ? 0x64006400u : magic;
                const f16x2 h0 = as_f16x2((qw & ml) | mg) + c1[c];             // k0,k4
                const f16x2 h1 = as_f16x2((qw & mh) | mg) * r16 + c2[c];       // k1,k5
                const f16x2 h2 = as_f16x2((q8 & ml) | mg) + c1[c];             // k2,k6
                const f16x2 h3 = as_f16x2((q8 & mh) | mg) * r16 + c2[c];       // k3,k7
                if constexpr (BF) {
                    // fp16 -> fp32 -> bf16 per pair (2 v_cvt_f32_f16 + 1 v_cvt_pk_bf16_f32; exact: integers in [-16, 15])
                    auto to_bf = [&](f16x2 hv) -> unsigned {
                        const bf16x2 o = {(bf16)(float)hv[0], (bf16)(float)hv[1]};
                        return __builtin_bit_cast(unsigned, o);
                    };
                    b01 = u32x2{to_bf(h0), to_bf(h1)};
                    b23 = u32x2{to_bf(h2), to_bf(h3)};
                } else {
                    b01 = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                    b23 = u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
                }
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    accg[rg][c] = Mma4<T>::run(xa01[rg], b01, accg[rg][c]);
                    accg[rg][c] = Mma4<T>::run(xa23[rg], b23, accg[rg][c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned sh = (c & 1) ? (sraw[c >> 1] >> 16) : (sraw[c >> 1] & 0xffffu);
            const float sc = DType<T>::to_f32(__builtin_bit_cast(T, (unsigned short)sh));
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int m = 0; m < MTR; ++m) {
                    acc[h][rg][c][m] = fmaf(sc, accg[rg][c][m], acc[h][rg][c][m]);
                }
        }
    }
    }
    // ---- reduce: row slots by shuffles, waves through LDS (one barrier), write ---------------------
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int m = 0; m < MTR; ++m)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[h][rg][c][m] = row_slot_sum<LN>(acc[h][rg][c][m]);
    constexpr int ROWS = RG * MTR;
    constexpr int E = ROWS * CT;                            // entries per half
    if (lane < LN) {
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int m = 0; m < MTR; ++m) {
                    f32x4 v = {acc[h][rg][0][m], acc[h][rg][1][m], acc[h][rg][2][m], acc[h][rg][3][m]};
                    *(f32x4*)(red + ((wave * H + h) * ROWS + rg * MTR + m) * CT + lane * 4) = v;
                }
    }
    __syncthreads();
    auto finish = [&](int e, float s0, float s1) {          // e = mm * CT + c, mm = rg * MTR + m
        const int mm = e / CT, c = e % CT;
        const int n = strip * CT + c, row = m0 + (mm / MTR) * 4 + (mm % MTR);
        if (n >= NH || row >= p.M) return;
        if constexpr (PAIR) {
            if (p.bias) { s0 += DType<T>::to_f32(((const T*)p.bias)[n]); s1 += DType<T>::to_f32(((const T*)p.bias)[n + p.pair_off]); }
            const float g = s0 / (1.f + __expf(-s0));            // silu on the fp32 sum
            ((T*)p.out)[(size_t)row * NH + n] = DType<T>::from_f32(g * s1);
        } else if (p.ksplit > 1) {
            p.partial[((size_t)blockIdx.y * p.M + row) * p.N + n] = s0;
        } else {
            if (p.bias) s0 += DType<T>::to_f32(((const T*)p.bias)[n]);
            ((T*)p.out)[(size_t)row * p.N + n] = DType<T>::from_f32(s0);
        }
    };
    // final cross-wave sum.  When E <= 64 one wave does it with P = 64 / E lanes per entry (each lane adds every
    // P-th wave's slab, then the P partial sums meet through shuffles) instead of E threads walking all W slabs.
    if constexpr (E <= 64 && (E & (E - 1)) == 0) {
        if (wave == 0) {
            constexpr int P = 64 / E;
            const int e = lane % E, part = lane / E;
            float s[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float t = 0.f;
                for (int w = part; w < W; w += P) t += red[(w * H + h) * E + e];
#pragma unroll
                for (int off = E; off < 64; off <<= 1) t += __shfl_xor(t, off, 64);
                s[h] = t;
            }
            if (part == 0) finish(e, s[0], s[H - 1]);
        }
    } else {
        for (int e = tid; e < E; e += blockDim.x) {
            float s[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float t = 0.f;
                for (int w = 0; w < W; ++w) t += red[(w * H + h) * E + e];
                s[h] = t;
            }
            finish(e, s[0], s[H - 1]);
        }
    }
}

// ---- magic-number field decode for the 3- and 8-bit fp16 matrix-core GEMV ------------------------------------------------------
// The generic kernel below extracts every field on its own (shift, mask, integer subtract, two conversions: ~5.3 VALU per
// weight; the 3-bit g32 decode launch is 912 VALU per wave and spends as long on them as on its loads).  For fp16 the same exact
// w - z comes out of packed arithmetic two fields at a time, like the 4-bit kernels do: (t & (7 << s) * 0x00010001) | 0x64006400 is
// the half2 (1024 + f_a * 2^s, 1024 + f_b * 2^s) for the two fields sitting at bit s of the two 16-bit halves of t, and ONE
// v_pk_fma_f16 by 2^-s with -(1024 * 2^-s + z) gives (f_a - z, f_b - z) exactly (s + bits <= 10: the field stays inside the
// mantissa; every intermediate is an integer below 2048).
//   8-bit: the word itself pairs (f0, f2) and, shifted by 8, (f1, f3): 1 shift + 2 v_and_or + 2 v_pk_add per 4 weights.
//   3-bit: 32 fields in 96 bits do not line up with the halves, but 16-bit WINDOWS of the bit stream at multiples of 15 bits do:
//          t_k = (stream >> 30k)[15:0] | (stream >> (30k + 15))[15:0] << 16 holds fields 10k..10k+4 in the low half and 10k+5..10k+9 in
//          the high half at the same bit positions 0, 3, 6, 9, 12 (9 and 12 are brought down to 3 and 6 by one shift of the whole
//          word); fields 30 and 31 are a fourth window pair.  <= 3 VALU per t_k, then 1 v_and_or + 1 v_pk op per pair: 46 VALU per
//          32 weights instead of ~170.
// The pairs put the unit's k values into matrix-core slots in the order ka(p), kb(p); x is brought into the same order with one
// v_perm per pair (shared by the lane's 4 columns).  bf16 has neither the mantissa (7 bits) nor packed arithmetic for this and keeps
// the field-by-field form, as do 2-bit layers.
// The masks and the magic number reach the expressions as OPAQUE register values (one-instruction asm definitions at kernel start,
// as in the 4-bit kernels): with literal operands hipcc splits (t & mask) | magic into v_and + v_or (VOP3 takes no literals on gfx9).
// The expressions themselves stay C on purpose.  A first version issued v_and_or_b32 from inline asm inside the loop and returned
// garbage / NaN in columns 1 and 2 of every lane at M = 2: with two rows of x the upper half of a 4x4x4 accumulator is dead, the
// register allocator reuses it for the next column's B fragments, and a VALU write hidden in asm is invisible to the hazard
// recognizer -- it landed BEFORE the in-flight MFMA's write of the same (dead) register, which then overwrote the fragment.
struct MagicConsts {
    unsigned magic, m0, m3, m6, m8;      // 0x64006400 (VGPR); 3-bit field masks at bits 0 / 3 / 6 of both halves, 8-bit mask (SGPRs)
    unsigned q0, q2, q4, q6;             // 2-bit field masks at bits 0 / 2 / 4 / 6 of both halves (SGPRs)
    __device__ __forceinline__ void init() {
        asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
        asm("s_mov_b32 %0, 0x00070007" : "=s"(m0));
        asm("s_mov_b32 %0, 0x00380038" : "=s"(m3));
        asm("s_mov_b32 %0, 0x01c001c0" : "=s"(m6));
        asm("s_mov_b32 %0, 0x00ff00ff" : "=s"(m8));
        asm("s_mov_b32 %0, 0x00030003" : "=s"(q0));
        asm("s_mov_b32 %0, 0x000c000c" : "=s"(q2));
        asm("s_mov_b32 %0, 0x00300030" : "=s"(q4));
        asm("s_mov_b32 %0, 0x00c000c0" : "=s"(q6));
    }
};
__device__ __forceinline__ unsigned f16x2_bits(f16x2 v) { return __builtin_bit_cast(unsigned, v); }

template <int BITS> struct MagicF16;
template <> struct MagicF16<8> {
    static constexpr int NP = 2;                                   // pairs per unit (4 values)
    static constexpr int ka(int p) { return p; }                  // (0, 2), (1, 3)
    static constexpr int kb(int p) { return p + 2; }
    f16x2 c1;
    __device__ __forceinline__ void setup(int z) { c1 = as_f16x2((unsigned)z * 0x00010001u + 0xE400E400u); }    // -(1024 + z), z <= 256
    __device__ __forceinline__ void pairs(const unsigned (&w)[1], const MagicConsts& k, unsigned (&bp)[NP]) const {
        bp[0] = f16x2_bits(as_f16x2((w[0] & k.m8) | k.magic) + c1);
        bp[1] = f16x2_bits(as_f16x2(((w[0] >> 8) & k.m8) | k.magic) + c1);
    }
};
// 2-bit (round 3): 16 values per word; v_perm spreads bytes 0 / 1 (values 0..7) and bytes 2 / 3 (values 8..15) over the halves of a register, so that
// the fields pair up as (f, f + 4) at bits 0 / 2 / 4 / 6 of both halves: 2 v_perm + 8 (v_and_or, packed add / fma) per 16 weights -- the field-by-field
// form this replaces made a 2-bit layer SLOWER than the 4-bit layer of twice the bytes (4096 x 11008, M = 1: 13.0 us against 8.4)
template <> struct MagicF16<2> {
    static constexpr int NP = 8;                                   // pairs per unit (16 values)
    static constexpr int ka(int p) { return p < 4 ? p : p + 4; }  // (0,4) (1,5) (2,6) (3,7) | (8,12) (9,13) (10,14) (11,15)
    static constexpr int kb(int p) { return ka(p) + 4; }
    f16x2 c1, c2, c4, c6;
    __device__ __forceinline__ void setup(int z) {                 // z <= 4
        c1 = as_f16x2((unsigned)z * 0x00010001u + 0xE400E400u);   // -(1024 + z)
        const f16x2 k768 = {(f16)768.f, (f16)768.f}, k960 = {(f16)960.f, (f16)960.f}, k1008 = {(f16)1008.f, (f16)1008.f};
        c2 = c1 + k768;                                            // -(256 + z), exact
        c4 = c1 + k960;                                            // -(64 + z)
        c6 = c1 + k1008;                                           // -(16 + z)
    }
    __device__ __forceinline__ void four(unsigned t, const MagicConsts& k, unsigned* bp) const {
        const f16x2 r4 = {(f16)0.25f, (f16)0.25f}, r16 = {(f16)0.0625f, (f16)0.0625f}, r64 = {(f16)0.015625f, (f16)0.015625f};
        bp[0] = f16x2_bits(as_f16x2((t & k.q0) | k.magic) + c1);
        bp[1] = f16x2_bits(as_f16x2((t & k.q2) | k.magic) * r4 + c2);
        bp[2] = f16x2_bits(as_f16x2((t & k.q4) | k.magic) * r16 + c4);
        bp[3] = f16x2_bits(as_f16x2((t & k.q6) | k.magic) * r64 + c6);
    }
    __device__ __forceinline__ void pairs(const unsigned (&w)[1], const MagicConsts& k, unsigned (&bp)[NP]) const {
        four(__builtin_amdgcn_perm(w[0], w[0], 0x0c010c00u), k, bp);          // byte 0 -> bits 0..7, byte 1 -> bits 16..23
        four(__builtin_amdgcn_perm(w[0], w[0], 0x0c030c02u), k, bp + 4);      // bytes 2, 3
    }
};
template <> struct MagicF16<3> {
    static constexpr int NP = 16;                                  // pairs per unit (32 values in 3 words)
    static constexpr int ka(int p) { return p < 15 ? 10 * (p / 5) + p % 5 : 30; }
    static constexpr int kb(int p) { return p < 15 ? ka(p) + 5 : 31; }
    f16x2 c1, c3, c6;
    __device__ __forceinline__ void setup(int z) {                 // z <= 8
        c1 = as_f16x2((unsigned)z * 0x00010001u + 0xE400E400u);   // -(1024 + z)
        const f16x2 k896 = {(f16)896.f, (f16)896.f}, k1008 = {(f16)1008.f, (f16)1008.f};
        c3 = c1 + k896;                                            // -(128 + z), exact
        c6 = c1 + k1008;                                           // -(16 + z), exact
    }
    __device__ __forceinline__ void five(unsigned t, const MagicConsts& k, unsigned* bp) const {
        const f16x2 r8 = {(f16)0.125f, (f16)0.125f}, r64 = {(f16)0.015625f, (f16)0.015625f};
        const unsigned t6 = t >> 6;
        bp[0] = f16x2_bits(as_f16x2((t & k.m0) | k.magic) + c1);
        bp[1] = f16x2_bits(as_f16x2((t & k.m3) | k.magic) * r8 + c3);
        bp[2] = f16x2_bits(as_f16x2((t & k.m6) | k.magic) * r64 + c6);
        bp[3] = f16x2_bits(as_f16x2((t6 & k.m3) | k.magic) * r8 + c3);
        bp[4] = f16x2_bits(as_f16x2((t6 & k.m6) | k.magic) * r64 + c6);
    }
    __device__ __forceinline__ void pairs(const unsigned (&w)[3], const MagicConsts& k, unsigned (&bp)[NP]) const {
        // 16-bit windows of the 96-bit stream at bits 0 / 15, 30 / 45, 60 / 75, 90 / 93 (low half / high half of t)
        const unsigned t0 = __builtin_amdgcn_perm(w[0] >> 15, w[0], 0x05040100u);
        const unsigned t1 = __builtin_amdgcn_perm(w[1] >> 13, __builtin_amdgcn_alignbit(w[1], w[0], 30), 0x05040100u);
        const unsigned t2 = __builtin_amdgcn_perm(w[2] >> 11, __builtin_amdgcn_alignbit(w[2], w[1], 28), 0x05040100u);
        const unsigned t3 = __builtin_amdgcn_perm(w[2] >> 29, w[2] >> 26, 0x05040100u);
        five(t0, k, bp);
        five(t1, k, bp + 5);
        five(t2, k, bp + 10);
        bp[15] = f16x2_bits(as_f16x2((t3 & k.m0) | k.magic) + c1);
    }
};
// x values of one unit (natural order, two per register) -> the register holding (x[ka(P)], x[kb(P)])
template <int BITS, int P, int NR>
__device__ __forceinline__ unsigned magic_x_pair(const unsigned (&xr)[NR]) {
    constexpr int a = MagicF16<BITS>::ka(P), b = MagicF16<BITS>::kb(P);
    constexpr unsigned sel = ((a & 1) ? 0x0302u : 0x0100u) | (((b & 1) ? 0x0706u : 0x0504u) << 16);
    return __builtin_amdgcn_perm(xr[b >> 1], xr[a >> 1], sel);
}

// ---- streamed matrix-core GEMV: weights by LDS DMA, up to 4 layers that share x in ONE launch ------------------
// Same lane decomposition and arithmetic as gemv_q4_f16_mfma_kernel (plain layers: no act-order, no fused epilogue,
// M <= 4), with two structural changes aimed at what bounds a one-shot decode launch -- bytes in flight and fixed cost
// per launch:
//  * the packed rows go global -> LDS by DMA (global_load_lds_dwordx4, nontemporal: 1 KiB per wave instruction, no VGPR
//    destination), each wave into its own U KiB region, and are read back by the SAME lanes (ds_read_b128, lane-linear:
//    conflict free) -- no barrier, the wave's own vmcnt orders DMA before read.  Bytes in flight per CU are bounded by the
//    160 KiB of LDS instead of the VGPR budget: U = 8 with 16 waves puts a whole 11008-row strip (88 KB) in flight from the
//    first cycle, where the register version walked it in 5.4 dependent iterations of 16 KB;  U = 4 with 8 waves stays
//    under 64 VGPRs, so the 688 strips of a 4096 x 11008 layer are all resident at once (3 workgroups per CU).
//  * the workgroup grid runs over the strips of up to four layers that read the same x (q/k/v, gate/up): one launch
//    boundary, one ramp and one drain for 25 / 45 MB instead of three / two (gptq_forward_multi).  The reference gets this
//    by concatenating the packed tensors (fused_llama_attn.py:171-186); here the checkpoint tensors stay where they are.
//  * K split (narrow layers / wide strips) is combined INSIDE the launch: every slice publishes its fp32 partials with
//    write-through (sc1) stores, drains them, takes a ticket; the slice that draws the last ticket sums all slices in
//    index order with sc1 loads (bit-reproducible: fixed order, no float atomics) and resets the ticket.  Correct for any
//    placement of the slices over XCDs / CUs.  Tickets live at the front of the workspace, which the caller hands over
//    zeroed once (gptq_mi355x.h).
template <int LN, int MT, int U, typename T, int WPS>
__global__ void __launch_bounds__(1024, WPS) gemv_q4_stream_kernel(GemvStreamParams p) {
    constexpr bool BF = std::is_same_v<T, bf16>;
    unsigned m_lo, m_hi, magic;
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WR = 64 / LN, CT = LN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    char* const wq = smem + (size_t)wave * (U * 1024);                        // this wave's DMA landing area
    const unsigned wq_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)wq);   // its LDS byte address (SGPR)
    float* const red = (float*)(smem + (size_t)W * (U * 1024));               // [W][MT][CT] cross-wave sums, + 1 word
    // logical block -> (strip over all layers, K slice); slices of one strip are adjacent logical ids (same XCD after the remap)
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int sidx = L / p.ksplit, ks = L - sidx * p.ksplit;
    int s = 0;
    while (s + 1 < p.nseg && sidx >= p.seg[s].blk_end) ++s;                   // wave-uniform (kernel arguments only)
    const GemvSeg& sg = p.seg[s];
    const int strip = sidx - (s ? p.seg[s - 1].blk_end : 0);
    const int N = sg.N;
    const int n0 = strip * CT + cl * 4;
    const bool col_ok = n0 < N;
    const int nload = col_ok ? n0 : 0;
    const int ub = ks * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    const T* xrow = (const T*)p.x + (size_t)min(lane & 3, p.M - 1) * p.K;     // A operand: lane i of a 4-lane group carries x row i
    const T* __restrict__ scales = (const T*)sg.scales;
    const unsigned* __restrict__ qweight = sg.qweight;
    const int zrow_words = N >> 3;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;
    const int gshift = p.gu_shift;

    f32x4 acc[MT];                 // [row of x] = the lane's 4 columns (whole-vector accesses only: a float[4][MT] was half-promoted, the rest sat in scratch)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int rows_per_iter = W * WR * U;
    for (int base = ub; base < ue; base += rows_per_iter) {
        const int u0 = base + (wave * WR + rs) * U;
        const int g = min(u0, ue - 1) >> gshift;
        // small L2-resident loads first (they return first), then the DMA burst
#if defined(GPTQ_STREAM_ABL) && (GPTQ_STREAM_ABL & 4)
        const u32x2 sraw = {0x14001400u + (unsigned)g, 0x14001400u};
        const unsigned zw = 0x77777777u;
#else
        const u32x2 sraw = *(const u32x2*)(scales + (size_t)g * N + nload);
        const unsigned zw = sg.qzeros[(size_t)g * zrow_words + (nload >> 3)];       // raw word: nothing is computed on loaded values up here
#endif
        u32x4 xr[U];
#if defined(GPTQ_STREAM_ABL) && (GPTQ_STREAM_ABL & 1)
#pragma unroll
        for (int j = 0; j < U; ++j) xr[j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (unsigned)u0};
#else
#pragma unroll
        for (int j = 0; j < U; ++j) xr[j] = *(const u32x4*)(xrow + (size_t)min(u0 + j, ue - 1) * 8);
#endif
        if (base != ub) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // WAR: last iteration's ds_reads are done
        // The burst goes out back to back: between the two fences there are only the DMAs and their address arithmetic.  Everything
        // that CONSUMES a loaded value (the zero-point shift, the x permutes) sits below the second fence in source order -- hipcc
        // had scheduled both into the burst, with an s_waitcnt for loads issued a few instructions earlier (a full memory latency)
        // behind the first DMA.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int ul = min(u0 + j, ue - 1);
            dma16_nt(qweight + (size_t)ul * N + nload, wq_lds + j * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned zw_ = zw >> ((nload & 7) * 4), s0_ = sraw[0], s1_ = sraw[1];
        f16x2 c1[4], c2[4];
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned z = (((zw_ >> (4 * c)) & 15u) + 1u) & zmask;
            c1[c] = as_f16x2(z * 0x00010001u + 0xE400E400u);        // -(1024+z)
            c2[c] = c1[c] + k960;                                   // -(64+z)
        }
        f32x4 accg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        // row j is consumed as soon as DMA j has landed: vmcnt retires in order and the U DMAs are the youngest VMEM
        // operations of the wave, so "at most U-1-j outstanding" means DMAs 0..j (and every older load) are complete
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] {
                 constexpr int j = J;
                 asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1 - j) : "memory");
                 const u32x4 qv = *(const u32x4*)(wq + j * 1024 + lane * 16);
#if defined(GPTQ_STREAM_ABL) && (GPTQ_STREAM_ABL & 2)
                 {
#pragma unroll
                     for (int c = 0; c < 4; ++c) accg[c][0] += as_f32((qv[c] ^ xr[j][c]) & 0x3fffffffu);
                     return;
                 }
#endif
                 const bool live = (u0 + j < ue);
                 const u32x4 t = xr[j];
                 u32x2 a01 = u32x2{__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u)};
                 u32x2 a23 = u32x2{__builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
                 if (!live) { a01 = u32x2{0u, 0u}; a23 = u32x2{0u, 0u}; }
#pragma unroll
                 for (int c = 0; c < 4; ++c) {
                     const unsigned qw = qv[c], q8 = qw >> 8;
                     const f16x2 h0 = as_f16x2((qw & m_lo) | magic) + c1[c];             // k0,k4
                     const f16x2 h1 = as_f16x2((qw & m_hi) | magic) * r16 + c2[c];       // k1,k5
                     const f16x2 h2 = as_f16x2((q8 & m_lo) | magic) + c1[c];             // k2,k6
                     const f16x2 h3 = as_f16x2((q8 & m_hi) | magic) * r16 + c2[c];       // k3,k7
                     u32x2 b01, b23;
                     if constexpr (BF) {
                         auto to_bf = [&](f16x2 hv) -> unsigned {
                             const bf16x2 o = {(bf16)(float)hv[0], (bf16)(float)hv[1]};
                             return __builtin_bit_cast(unsigned, o);
                         };
                         b01 = u32x2{to_bf(h0), to_bf(h1)};
                         b23 = u32x2{to_bf(h2), to_bf(h3)};
                     } else {
                         b01 = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                         b23 = u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
                     }
                     accg[c] = Mma4<T>::run(a01, b01, accg[c]);
                     accg[c] = Mma4<T>::run(a23, b23, accg[c]);
                 }
             }()),
             ...);
        }(std::make_integer_sequence<int, U>{});
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned sw = (c >> 1) ? s1_ : s0_;
            const unsigned sh = (c & 1) ? (sw >> 16) : (sw & 0xffffu);
            const float sc = DType<T>::to_f32(__builtin_bit_cast(T, (unsigned short)sh));
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][c] = fmaf(sc, accg[c][m], acc[m][c]);
        }
    }
#if defined(GPTQ_STREAM_ABL) && (GPTQ_STREAM_ABL & 8)
    if (acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] == 1.2345f) ((T*)sg.out)[n0] = DType<T>::from_f32(acc[0][0]);
    return;
#endif
    stream_epilogue<LN, MT, T>(acc, p, sg, strip, sidx, ks, N, red);
}

// ---- the streamed kernel for 3- and 8-bit fp16 layers (BASELINE config 5) ----------------------------------------------------------------
// The structure of gemv_q4_stream_kernel (weights by LDS DMA into per-wave landing areas, up to four layers that share x in one launch,
// K slices combined through granules) with the packing units and the magic-number decode of gemv_mfma_generic_kernel: a unit is one word
// of 4 values (8-bit) or three words of 32 values (3-bit), a lane owns 4 adjacent columns and U consecutive units of ONE group, every
// word of a unit is its own 1 KiB wave DMA (row (unit * UW + w) of the packed matrix), and unit j is consumed as soon as its UW DMAs
// have landed.  Same values as the register kernel, bit for bit (same pairs, same matrix-core order, fp32 group sums times the scale).
template <int BITS, int LN, int MT, int U>
__global__ void __launch_bounds__(1024, 4) gemv_qx_stream_kernel(GemvStreamParams p) {
    static_assert(BITS == 2 || BITS == 3 || BITS == 8, "4-bit layers have their own kernel");
    using T = f16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals, WR = 64 / LN, CT = LN * 4;
    constexpr int XV = KPU / 8;                                               // 16-byte x pieces per unit (8-bit: half a piece)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    char* const wq = smem + (size_t)wave * (U * UW * 1024);                   // this wave's DMA landing area
    const unsigned wq_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)wq);
    float* const red = (float*)(smem + (size_t)W * (U * UW * 1024));          // [W][MT][CT] cross-wave sums, + 1 word
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int sidx = L / p.ksplit, ks = L - sidx * p.ksplit;
    int s = 0;
    while (s + 1 < p.nseg && sidx >= p.seg[s].blk_end) ++s;                   // wave-uniform (kernel arguments only)
    const GemvSeg& sg = p.seg[s];
    const int strip = sidx - (s ? p.seg[s - 1].blk_end : 0);
    const int N = sg.N;
    const int n0 = strip * CT + cl * 4;
    const bool col_ok = n0 < N;
    const int nload = col_ok ? n0 : 0;
    const int ub = ks * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    const T* xrow = (const T*)p.x + (size_t)min(lane & 3, p.M - 1) * p.K;     // A operand: lane i of a 4-lane group carries x row i
    const T* __restrict__ scales = (const T*)sg.scales;
    const unsigned* __restrict__ qweight = sg.qweight;
    const int zrow_words = N / 32 * BITS;
    const unsigned zbit = (unsigned)BITS * (unsigned)nload;                    // the lane's 4 zero-points: 4 * BITS bits from here (one or two words)
    const unsigned zwi = zbit >> 5, zsh = zbit & 31;
    const bool z2 = zsh + 4u * BITS > 32u;
    const int gshift = p.gu_shift;
    constexpr unsigned maxq = (1u << BITS) - 1u;

    f32x4 acc[MT];                 // [row of x] = the lane's 4 columns (whole-vector accesses only: a float[4][MT] was half-promoted, the rest sat in scratch)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    MagicConsts mk;
    mk.init();

    const int rows_per_iter = W * WR * U;
    for (int base = ub; base < ue; base += rows_per_iter) {
        const int u0 = base + (wave * WR + rs) * U;
        const int g = min(u0, ue - 1) >> gshift;
        // small L2-resident loads first (they return first), then the DMA burst; nothing is computed on loaded values up here
        const u32x2 sraw = *(const u32x2*)(scales + (size_t)g * N + nload);
        const unsigned* zrow = sg.qzeros + (size_t)g * zrow_words;
        const unsigned zlo = zrow[zwi];
        const unsigned zhi = z2 ? zrow[zwi + 1] : 0u;
        unsigned xr[U][KPU / 2];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const T* xp = xrow + (size_t)min(u0 + j, ue - 1) * KPU;
            if constexpr (KPU == 4) {
                const u32x2 t = *(const u32x2*)xp;
                xr[j][0] = t[0]; xr[j][1] = t[1];
            } else {
#pragma unroll
                for (int v = 0; v < XV; ++v) {
                    const u32x4 t = *(const u32x4*)(xp + 8 * v);
                    xr[j][4 * v] = t[0]; xr[j][4 * v + 1] = t[1]; xr[j][4 * v + 2] = t[2]; xr[j][4 * v + 3] = t[3];
                }
            }
        }
        if (base != ub) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // WAR: last iteration's ds_reads are done
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int ul = min(u0 + j, ue - 1);
#pragma unroll
            for (int w = 0; w < UW; ++w) dma16_nt(qweight + (size_t)(ul * UW + w) * N + nload, wq_lds + (j * UW + w) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long zv = (((unsigned long long)zhi << 32) | zlo) >> zsh;
        MagicF16<BITS> mg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = (int)((unsigned)(zv >> (BITS * c)) & maxq) + 1;
            mg[c].setup((p.zero_mode == GPTQ_ZERO_WRAP) ? (f & (int)maxq) : f);
        }
        f32x4 accg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // unit j is consumed as soon as its UW DMAs have landed (vmcnt retires in order; the U * UW DMAs are the wave's youngest VMEM operations)
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] {
                 constexpr int j = J;
                 asm volatile("s_waitcnt vmcnt(%0)" ::"n"((U - 1 - j) * UW) : "memory");
                 u32x4 qv[UW];
#pragma unroll
                 for (int w = 0; w < UW; ++w) qv[w] = *(const u32x4*)(wq + (j * UW + w) * 1024 + lane * 16);
                 const bool live = (u0 + j < ue);
                 unsigned xa[KPU / 2];                                        // x in the slot order of the pairs, shared by the 4 columns
                 [&]<int... P>(std::integer_sequence<int, P...>) {
                     ((xa[P] = live ? magic_x_pair<BITS, P>(xr[j]) : 0u), ...);
                 }(std::make_integer_sequence<int, KPU / 2>{});
#pragma unroll
                 for (int c = 0; c < 4; ++c) {
                     unsigned wds[UW], bp[KPU / 2];
#pragma unroll
                     for (int w = 0; w < UW; ++w) wds[w] = qv[w][c];
                     mg[c].pairs(wds, mk, bp);
#pragma unroll
                     for (int Q = 0; Q < KPU / 4; ++Q)
                         accg[c] = Mma4<T>::run(u32x2{xa[2 * Q], xa[2 * Q + 1]}, u32x2{bp[2 * Q], bp[2 * Q + 1]}, accg[c]);
                 }
             }()),
             ...);
        }(std::make_integer_sequence<int, U>{});
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned sh = (c & 1) ? (sraw[c >> 1] >> 16) : (sraw[c >> 1] & 0xffffu);
            const float sc = DType<T>::to_f32(__builtin_bit_cast(T, (unsigned short)sh));
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][c] = fmaf(sc, accg[c][m], acc[m][c]);
        }
    }
    stream_epilogue<LN, MT, T>(acc, p, sg, strip, sidx, ks, N, red);
}

// ---- matrix-core GEMV, any bits, fp16 / bf16 -----------------------------------------------------
// The same structure for the other packings (2/3/8-bit, and 4-bit with bf16): a lane owns 4 columns and U consecutive
// packing units (1 word = 16/8/4 values, or 3 words = 32 values for 3-bit) of one group; fields are extracted with
// v_bfe, w - z is formed in integers and converted exactly to T (|w - z| <= 256 fits both fp16 and bf16), 4 values of
// the lane's own column feed one 4x4x4 MFMA against the matching 4 x values of up to 4 x rows (natural k order: no x
// permutation).  out = sum_g s_g * (sum_{k in g} x_k (w_k - z_g)), fp32 sums.
// MAGIC (3- / 8-bit, fp16): the packed magic-number decode above instead of the field-by-field one; same values, bit for bit.
// MT = 8 (MAGIC only): 5..8 rows of x in ONE pass over the weights -- a second A operand (rows 4..7) and a second MFMA per fragment, as the
// 4-bit kernel does; 8 waves at most (<= 256 VGPRs).  Without it 5..8 rows are two passes (blockIdx.z) that unpack every word twice.
template <int BITS, typename T, int LN, int MT, int U, bool MAGIC = false>
__global__ void __launch_bounds__(MT == 8 ? 512 : 1024, MT == 8 ? 2 : 4) gemv_mfma_generic_kernel(GemvParams p) {
    static_assert(!MAGIC || BITS == 2 || BITS == 3 || BITS == 8, "magic-number decode: 2- / 3- / 8-bit (4-bit has its own kernels)");
    static_assert(MT <= 4 || (MT == 8 && MAGIC && std::is_same_v<T, f16>), "8 rows per pass: fp16 magic-number variants only");
    constexpr int RP = MT == 8 ? 8 : 4;                 // rows of x per pass (blockIdx.z)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;
    constexpr int UW = Pack<BITS>::words, KPU = Pack<BITS>::vals, WR = 64 / LN, CT = LN * 4;
    constexpr int XV = KPU / 8;                         // 16-byte x pieces per unit (8-bit: half a piece)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    const int cl = lane % LN, rs = lane / LN;
    const int strip = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = strip * CT + cl * 4;
    const bool col_ok = n0 < p.N;
    const int nload = col_ok ? n0 : 0;
    const int m0 = blockIdx.z * RP;
    const int ub = blockIdx.y * p.units_per_split;
    const int ue = min(ub + p.units_per_split, p.units_total);
    const T* __restrict__ xrow = (const T*)p.x + (size_t)min(m0 + (lane & 3), p.M - 1) * p.K;
    const T* __restrict__ xrow2 = (const T*)p.x + (size_t)min(m0 + 4 + (lane & 3), p.M - 1) * p.K;     // MT = 8: rows 4..7
    const T* __restrict__ scales = (const T*)p.scales;
    const int zrow_words = p.N / 32 * BITS;
    const int gunits = p.group_size / KPU;              // units per group (>= 1, the launcher checks divisibility)

    float acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;

    MagicConsts mk;
    if constexpr (MAGIC) mk.init();
    const int rows_per_iter = W * WR * U;
    for (int base = ub; base < ue; base += rows_per_iter) {
        const int u0 = base + (wave * WR + rs) * U;
        const int g = min(u0, ue - 1) / gunits;
        const u32x2 sraw = *(const u32x2*)(scales + (size_t)g * p.N + nload);
        int z[4];
        zero_points4(p.qzeros + (size_t)g * zrow_words, nload, BITS, p.zero_mode, z);
        // x: KPU values per unit (8-byte piece for 8-bit, 1/2/4 16-byte pieces for 4/2/3-bit)
        unsigned xr[U][KPU / 2];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const T* xp = xrow + (size_t)min(u0 + j, ue - 1) * KPU;
            if constexpr (KPU == 4) {
                const u32x2 t = *(const u32x2*)xp;
                xr[j][0] = t[0]; xr[j][1] = t[1];
            } else {
#pragma unroll
                for (int v = 0; v < XV; ++v) {
                    const u32x4 t = *(const u32x4*)(xp + 8 * v);
                    xr[j][4 * v] = t[0]; xr[j][4 * v + 1] = t[1]; xr[j][4 * v + 2] = t[2]; xr[j][4 * v + 3] = t[3];
                }
            }
        }
        unsigned xr2[MT == 8 ? U : 1][KPU / 2];
        if constexpr (MT == 8) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const T* xp = xrow2 + (size_t)min(u0 + j, ue - 1) * KPU;
                if constexpr (KPU == 4) {
                    const u32x2 t = *(const u32x2*)xp;
                    xr2[j][0] = t[0]; xr2[j][1] = t[1];
                } else {
#pragma unroll
                    for (int v = 0; v < XV; ++v) {
                        const u32x4 t = *(const u32x4*)(xp + 8 * v);
                        xr2[j][4 * v] = t[0]; xr2[j][4 * v + 1] = t[1]; xr2[j][4 * v + 2] = t[2]; xr2[j][4 * v + 3] = t[3];
                    }
                }
            }
        }
        u32x4 q[U][UW];
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int w = 0; w < UW; ++w)
                q[j][w] = __builtin_nontemporal_load((const u32x4*)(p.qweight + (size_t)(min(u0 + j, ue - 1) * UW + w) * p.N + nload));

        f32x4 accg[4], accg2[MT == 8 ? 4 : 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) accg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (MT == 8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) accg2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (MAGIC) {
            MagicF16<BITS> mg[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) mg[c].setup(z[c]);
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool live = (u0 + j < ue);
                unsigned xa[KPU / 2], xa2[MT == 8 ? KPU / 2 : 1];   // x in the slot order of the pairs, shared by the 4 columns
                [&]<int... P>(std::integer_sequence<int, P...>) {
                    ((xa[P] = live ? magic_x_pair<BITS, P>(xr[j]) : 0u), ...);
                    if constexpr (MT == 8) ((xa2[P] = live ? magic_x_pair<BITS, P>(xr2[j]) : 0u), ...);
                }(std::make_integer_sequence<int, KPU / 2>{});
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned wds[UW], bp[KPU / 2];
#pragma unroll
                    for (int w = 0; w < UW; ++w) wds[w] = q[j][w][c];
                    mg[c].pairs(wds, mk, bp);
                    if constexpr (std::is_same_v<T, bf16>) {          // bf16 (round 3): the exact fp16 w - z, each pair converted once (|w - z| <= 256 is exact in bf16)
#pragma unroll
                        for (int i = 0; i < KPU / 2; ++i) {
                            const f16x2 hv = as_f16x2(bp[i]);
                            const bf16x2 o = {(bf16)(float)hv[0], (bf16)(float)hv[1]};
                            bp[i] = __builtin_bit_cast(unsigned, o);
                        }
                    }
#pragma unroll
                    for (int Q = 0; Q < KPU / 4; ++Q) {
                        accg[c] = Mma4<T>::run(u32x2{xa[2 * Q], xa[2 * Q + 1]}, u32x2{bp[2 * Q], bp[2 * Q + 1]}, accg[c]);
                        if constexpr (MT == 8)
                            accg2[c] = Mma4<T>::run(u32x2{xa2[2 * Q], xa2[2 * Q + 1]}, u32x2{bp[2 * Q], bp[2 * Q + 1]}, accg2[c]);
                    }
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool live = (u0 + j < ue);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned wds[UW];
#pragma unroll
                for (int w = 0; w < UW; ++w) wds[w] = q[j][w][c];
                [&]<int... Q>(std::integer_sequence<int, Q...>) {          // Q = quad of 4 consecutive k inside the unit
                    (([&] {
                         const unsigned short e0 = Mma4<T>::bits_of_int((int)unit_field<BITS, 4 * Q + 0>(wds) - z[c]);
                         const unsigned short e1 = Mma4<T>::bits_of_int((int)unit_field<BITS, 4 * Q + 1>(wds) - z[c]);
                         const unsigned short e2 = Mma4<T>::bits_of_int((int)unit_field<BITS, 4 * Q + 2>(wds) - z[c]);
                         const unsigned short e3 = Mma4<T>::bits_of_int((int)unit_field<BITS, 4 * Q + 3>(wds) - z[c]);
                         const u32x2 b = {(unsigned)e0 | ((unsigned)e1 << 16), (unsigned)e2 | ((unsigned)e3 << 16)};
                         u32x2 a = {xr[j][2 * Q], xr[j][2 * Q + 1]};
                         if (!live) a = u32x2{0u, 0u};
                         accg[c] = Mma4<T>::run(a, b, accg[c]);
                     }()),
                     ...);
                }(std::make_integer_sequence<int, KPU / 4>{});
            }
        }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned sh = (c & 1) ? (sraw[c >> 1] >> 16) : (sraw[c >> 1] & 0xffffu);
            const float sc = DType<T>::to_f32(__builtin_bit_cast(T, (unsigned short)sh));
#pragma unroll
            for (int m = 0; m < (MT == 8 ? 4 : MT); ++m) acc[c][m] = fmaf(sc, accg[c][m], acc[c][m]);
            if constexpr (MT == 8) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[c][4 + m] = fmaf(sc, accg2[c][m], acc[c][4 + m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c][m] = row_slot_sum<LN>(acc[c][m]);
    if (lane < LN) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 v = {acc[0][m], acc[1][m], acc[2][m], acc[3][m]};
            *(f32x4*)(red + (wave * MT + m) * CT + lane * 4) = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < MT * CT; i += blockDim.x) {
        const int m = i / CT, c = i % CT;
        const int n = strip * CT + c, row = m0 + m;
        if (n >= p.N || row >= p.M) continue;
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += red[(w * MT + m) * CT + c];
        if (p.ksplit > 1) {
            p.partial[((size_t)blockIdx.y * p.M + row) * p.N + n] = s;
        } else {
            if (p.bias) s += DType<T>::to_f32(((const T*)p.bias)[n]);
            ((T*)p.out)[(size_t)row * p.N + n] = DType<T>::from_f32(s);
        }
    }
}

// ---- second pass for ksplit > 1: out = sum_s partial[s] (+bias), fixed order ----------------
template <typename T>
__global__ void __launch_bounds__(256) gemv_reduce_kernel(const float* __restrict__ partial, const T* __restrict__ bias,
                                                          T* __restrict__ out, int S, int M, int N) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += partial[(size_t)k * total + i];
        if (bias) s += DType<T>::to_f32(bias[i % N]);
        out[i] = DType<T>::from_f32(s);
    }
}

// ---- host side: shape heuristic + dispatch ---------------------------------------------------
static int pick_mt(int M) { return M >= 8 ? 8 : (M >= 4 ? 4 : (M >= 2 ? 2 : 1)); }

static GemvPlan plan_gemv_n(const gptq_layer_t& L, int M, const gptq_tuning_t* tune, int N_cols);

GemvPlan plan_gemv(const gptq_layer_t& L, int M, const gptq_tuning_t* tune) {
    if (L.epilogue == GPTQ_EPI_SILU_MUL) {
        // fused epilogue: the workgroup grid covers the N/2 output columns; only the matrix-core kernel implements it
        GemvPlan pl = plan_gemv_n(L, M, tune, L.N / 2);
        if (pl.mfma && pl.ksplit == 1 && pl.ln == 4 && M <= 4) {      // (5..8 rows: the unfused call -- the 8-row PAIR forms spilled and lost to it)
            pl.pair = true;
            if (pl.u > 2) pl.u = 2;
            pl.lds_bytes *= 2;
            return pl;
        }
    }
    // act-order with 2+ rows of x: the in-kernel gather through perm[] (x row in LDS, cooperative per-wave gather) is built for one
    // row; for MT = 2 / 4 it measured 22-40 us on the 23 MB layers against 9.5-11.5 us for the plain kernel (tools/act_batch.py).
    // x is permuted ONCE by the column-permute pre-pass (a 2.6 us launch for these sizes) and the plain matrix-core kernel runs on the
    // re-sequenced rows: 4096x11008 M = 2 / 4: 22.2 / 28.9 -> 12.1 / 14.1 us.
    // (also for ONE row when K is too long for the x row in LDS: 28672x1024 15.8 us on the LDS-staged fallback, 12.1 this way)
    // 2/3/8-bit act-order layers (round 2): the same, from ONE row on -- their matrix-core kernel has no in-kernel gather at all, so the
    // alternative is the fp32 generic kernel (tools/cliff_scan.py, us, M = 1 / 4: int3 g32 11008x4096 18.5 / 30.2, int8 23.1 / 39.7 against
    // 9.6 / 12.7 and 16.3 / 19.3 for the plain layer + the 2.6 us pre-pass)
    const bool q4_rows = L.bits == 4 && (M >= 2 || L.K > 24576);
    if (L.g_idx != nullptr && L.qweight_seq != nullptr && L.perm != nullptr && (q4_rows || L.bits != 4) && L.epilogue == GPTQ_EPI_NONE &&
        (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16) && (!tune || tune->path == 0 || (tune->path == 5 && L.bits != 4))) {
        gptq_layer_t P = L;
        P.g_idx = nullptr; P.perm = nullptr; P.qweight = L.qweight_seq; P.qweight_seq = nullptr;
        GemvPlan pp = plan_gemv_n(P, M, tune, L.N);
        if (L.bits == 4 ? pp.mfma : pp.mfmag) {
            pp.pair = false;
            pp.xperm = true;
            pp.xperm_bytes = ((size_t)M * L.K * 2 + 255) / 256 * 256;
            pp.workspace_bytes += pp.xperm_bytes;
            return pp;
        }
    }
    GemvPlan pl = plan_gemv_n(L, M, tune, L.N);
    pl.pair = false;
    return pl;
}

static GemvPlan plan_gemv_n(const gptq_layer_t& L, int M, const gptq_tuning_t* tune, int N_cols) {
    GemvPlan pl{};
    const int kpu = unit_vals(L.bits);
    const int path = tune ? tune->path : 0;
    pl.units_total = L.K / kpu;
    const bool uniform_groups = (L.group_size % kpu == 0);
    const bool seq = (L.g_idx == nullptr) || (L.qweight_seq != nullptr && L.perm != nullptr);
    pl.perk = !(uniform_groups && seq);
    pl.use_seq = (L.g_idx != nullptr) && !pl.perk;
    // matrix-core kernels (no LDS staging): power-of-two packed rows per group
    const int gu = L.group_size / 8;
    const bool q4_16 = L.bits == 4 && !pl.perk && path != 1 && (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16);
    const bool pow2_groups = q4_16 && gu > 0 && (gu & (gu - 1)) == 0;
    const bool ln_ok = !(tune && tune->lanes_n && tune->lanes_n != 4);
    // default for 4-bit fp16 / bf16 layers; act-order layers (x gathered through perm) and bf16 only with 16-column strips
    pl.mfma = pow2_groups && (path == 0 || path == 5) && (!pl.use_seq || (ln_ok && L.K <= 24576)) &&
              (L.dtype == GPTQ_F16 || ln_ok);
    // the other packings (and 4-bit bf16): matrix-core kernel with integer field extraction; 16-column strips only
    pl.mfmag = !pl.mfma && (path == 0 || path == 5) && !pl.perk && !pl.use_seq && ln_ok &&
               (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16) && !(L.bits == 4 && L.dtype == GPTQ_F16);
    if (pl.mfma) {                 // the matrix core handles 4 rows of x per pass, whatever M is
        pl.mt = M >= 5 ? 8 : (M >= 3 ? 4 : M);         // 8 = two groups of 4 rows in one pass
        pl.mtiles = M >= 5 ? (M + 7) / 8 : 1;
    } else if (pl.mfmag) {
        // 3- / 8-bit fp16 (magic-number decode): 5..8 rows in one pass (two matrix-core sets per fragment), else 4 rows per pass
        const bool magic_ok = L.dtype == GPTQ_F16 && (L.bits == 3 || L.bits == 8) && !(tune && tune->reserved[1] == 1);      // (2-bit: 4 rows per pass)
        // ... on layers of at most 256 strips (tools/nonq4_paths.py, profiles/r02_nonq4_paths.log, us at M = 8, two passes -> one: int3 11008x4096
        // 22.8 -> 16.8, 4096x4096 8.6 -> 8.9; on 4096x11008 -- 688 strips, three 8-wave workgroups per CU at 137 VGPRs -- 17.7 -> 21.6: stays two passes)
        if (magic_ok && M >= 5 && (N_cols <= 4096 || (tune && tune->path == 5))) {
            pl.mt = 8;
            pl.mtiles = (M + 7) / 8;
        } else {
            pl.mt = M >= 3 ? 4 : M;
            pl.mtiles = (M + 3) / 4;
        }
    } else {
        pl.mt = pick_mt(M);
        if (pl.mt > 4) pl.mt = 4;
        if (L.bits == 3) pl.mt = pl.perk ? 1 : (pl.mt > 2 ? 2 : pl.mt);   // 32 fields per unit: more rows of x spill
        pl.mtiles = (M + pl.mt - 1) / pl.mt;
    }

    int ln = (tune && tune->lanes_n) ? tune->lanes_n : 0;
    if (!ln) {
        if (pl.mfma || pl.mfmag) {
            // measured (tools/gemvlab, tools/membench): without a K split the 16-column strip wins on every
            // Llama shape -- a second (reduce) launch costs more than the 64-byte row segments do
            ln = 4;
            // ... for the Llama-7B widths.  Much wider fp16 layers have enough columns for wider strips AND enough workgroups:
            // from 12288 columns up, the widest strip (64, then 32 columns: 256- / 128-byte row segments) that still leaves
            // >= 160 workgroups (tools/gemv_sweep.py, us per launch, 16-column strips -> chosen: 4096x12288 10.5 -> 9.0,
            // 5120x13824 16.8 -> 11.0, 4096x14336 11.5 -> 9.2, 8192x28672 36.5 -> 27.8 = 4.4 TB/s; confirmed inside the
            // decode graph by the fused [q|k|v] launch: 2625 -> 2730 GB/s for the fused stack); 32-column strips when they
            // tile the chip exactly (N = 8192: 13.1 -> 12.1 us).  NOT for 11008 columns: 64-column strips look 5 % faster
            // back to back (9.1 vs 9.6 us) but are 11 % slower between the other layers of a decoder block (10.6 us), and
            // not for bf16, whose conversion work needs every CU (4096x11008: 12.4 -> 14.6 us with 172 workgroups).
            if (pl.mfma && L.dtype == GPTQ_F16 && !pl.use_seq && L.epilogue == GPTQ_EPI_NONE && M <= 4) {
                if (N_cols >= 12288) {
                    for (int cand : {16, 8}) {
                        const int strips = (N_cols + cand * 4 - 1) / (cand * 4);
                        if (N_cols % (cand * 4) == 0 && strips * pl.mtiles >= 160) { ln = cand; break; }
                    }
                } else if (N_cols % 8192 == 0) {                  // M <= 4 (3584x8192, us: M = 2 8.6 -> 7.5, M = 4 9.9 -> 8.7)
                    ln = 8;
                }
            }
        } else {
            ln = 4;   // widest strip that still gives >= 256 workgroups; else the narrowest (16 columns)
            for (int cand : {16, 8}) {
                if (cand == 8 && L.dtype != GPTQ_F32) continue;       // 32-column strips: instantiated for fp32 layers only
                const int strips = (N_cols + cand * 4 - 1) / (cand * 4);
                if (strips * pl.mtiles >= 256) { ln = cand; break; }
            }
        }
    }
    pl.ln = ln;
    const int wr = 64 / ln;
    pl.strips = (N_cols + ln * 4 - 1) / (ln * 4);
    int ks = (tune && tune->ksplit) ? tune->ksplit : 0;
    if (!ks) {
        ks = 1;
        while (pl.strips * pl.mtiles * ks < 192 && pl.units_total / (ks * 2) >= wr * 4) ks *= 2;
    }
    if (ks > pl.units_total) ks = pl.units_total;
    pl.ksplit = ks;
    pl.units_per_split = (pl.units_total + ks - 1) / ks;
    int waves = (tune && tune->waves) ? tune->waves : 0;
    const bool act_m1 = pl.mfma && pl.use_seq && M == 1 && L.epilogue != GPTQ_EPI_SILU_MUL;
    if (!waves) {
        waves = (pl.units_per_split + wr - 1) / wr;   // one unit per lane if possible
        if (waves > 16) waves = 16;
        // act-order decode: 8 waves with 2-4 rows per lane beat 16 waves with 1-2 on all three Llama-7B shapes (tools/act_ab.py,
        // profiles/r02_act_order_decode_ab.log: fp16 5.9 / 13.5 / 12.0 -> 5.1 / 9.8 / 9.6 us, bf16 6.5 / 15.5 / 14.8 -> 5.4 / 11.1 / 11.1):
        // two workgroups share a CU, so one gathers x through perm[] while the other's loads are in flight
        if (act_m1 && waves > 8) waves = 8;
        // 3-bit fp16 (magic-number decode), one row of x: 8 waves (two workgroups per CU) also on long-K layers -- tools/magic_ab.py sweep,
        // profiles/r02_magic_decode_ab.log: 11008x4096 g32 10.6 us with 16 waves x 2 passes, 9.6 with 8 x 3; 4096x4096 / 4096x11008 already run 8
        if (pl.mfmag && L.bits == 3 && L.dtype == GPTQ_F16 && M == 1 && !(tune && tune->reserved[1] == 1) && waves > 8) waves = 8;
        if (waves < 1) waves = 1;
    }
    if (pl.mfmag && pl.mt == 8 && waves > 8) waves = 8;      // 8 rows per pass: 512-thread workgroups (register budget), also when forced
    pl.waves = waves;
    const int rows_per_iter = wr * waves;
    const size_t rbytes = (size_t)waves * pl.mt * ln * 4 * sizeof(float);
    pl.workspace_bytes = ks > 1 ? (size_t)ks * M * L.N * sizeof(float) : 0;
    pl.u = 1;
    // 3- / 8-bit fp16: packed magic-number decode (tuning.reserved[1] = 1 keeps the field-by-field form, for A/B runs)
    // bf16 (tools/bf16_magic_ab.py, profiles/r03_bf16_magic_decode_ab.log): the packed decode + one conversion per pair wins for 2- and 3-bit (int2 4096x11008
    // 12.7 -> 10.6 us, int3 11008x4096 15.4 -> 13.4) and loses for 8-bit (18.1 -> 19.6: four byte fields per word are cheap to extract one by one)
    pl.magic = pl.mfmag && (L.dtype == GPTQ_F16 || (L.dtype == GPTQ_BF16 && L.bits != 8)) && (L.bits == 2 || L.bits == 3 || L.bits == 8) &&
               !(tune && tune->reserved[1] == 1);
    if (pl.mfmag) {
        const int gunits = L.group_size / kpu;
        const int per_lane = (pl.units_per_split + rows_per_iter - 1) / rows_per_iter;
        const int ucap = L.bits == 3 ? 1 : (L.bits == 2 ? 2 : 4);      // register budget: U * words and U * x values per lane
        int u = 1;
        for (int cand : {4, 2})
            if (cand <= ucap && cand <= per_lane && gunits % cand == 0 && pl.units_per_split % cand == 0) { u = cand; break; }
        pl.u = u;
        pl.chunk_units = pl.units_per_split;
        pl.lds_bytes = rbytes;
        return pl;
    }
    if (pl.mfma) {
        // consecutive packed rows per lane and iteration: same group, so one (scales, zeros) fetch serves them
        const int per_lane = (pl.units_per_split + rows_per_iter - 1) / rows_per_iter;
        int want = pl.units_per_split >= 1024 ? 1 : 2;             // long K: more, shorter iterations pipeline better
        // more than one workgroup per CU only fits with <= 64 VGPRs: U = 1 once 3+ rows of x are carried
        if (pl.mfma && pl.mt == 4 && (long)pl.strips * pl.mtiles > 256) want = 1;
        // act-order: the x gather is a dependent round trip per iteration -> few iterations; bf16 pays conversions per row, U = 2
        if (pl.use_seq) want = act_m1 ? ((L.dtype == GPTQ_BF16 && L.K <= 8192) ? 2 : 4) : 2;
        int u = 1;
        while (u * 2 <= per_lane && u * 2 <= gu && u * 2 <= 4 && 
               pl.units_per_split % (u * 2) == 0 && (tune && tune->reserved[0] > 0 ? u * 2 <= tune->reserved[0] : u * 2 <= want))
            u *= 2;
        pl.u = u;
        pl.chunk_units = pl.units_per_split;
        pl.lds_bytes = rbytes;
        if (pl.mfma && pl.use_seq && pl.mt == 1) pl.lds_bytes += (size_t)L.K * 2;   // whole x row for the LDS gather
        return pl;
    }
    // gemv_generic_kernel: x tile in LDS per K-chunk, kept <= 64 KiB
    auto lds_for = [&](int cu) -> size_t {
        return (size_t)pl.mt * cu * kpu * 4;
    };
    int cu = pl.units_per_split;
    while (cu > rows_per_iter && lds_for(cu) > 64 * 1024) cu = ((cu / 2 + rows_per_iter - 1) / rows_per_iter) * rows_per_iter;
    pl.chunk_units = cu;
    const size_t xbytes = lds_for(cu);
    pl.lds_bytes = xbytes > rbytes ? xbytes : rbytes;
    return pl;
}

template <int LN, int MT, typename T>
static hipError_t launch_mfma_u(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    dim3 grid(pl.strips, pl.ksplit, pl.mtiles), block(pl.waves * 64);
    if (pl.use_seq) {
        if constexpr (LN == 4) {
            if (pl.pair) {
                if constexpr (MT <= 4) {
                    if (pl.u == 1) hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 1, true, true, T>), grid, block, pl.lds_bytes, st, p);
                    else hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 2, true, true, T>), grid, block, pl.lds_bytes, st, p);
                } else {
                    return hipErrorInvalidValue;
                }
            } else if (pl.u == 1) {
                hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 1, false, true, T>), grid, block, pl.lds_bytes, st, p);
            } else if (pl.u == 2) {
                hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 2, false, true, T>), grid, block, pl.lds_bytes, st, p);
            } else if (pl.u == 4) {
                hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 4, false, true, T>), grid, block, pl.lds_bytes, st, p);
            } else {
                return hipErrorInvalidValue;
            }
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (pl.pair) {
        if constexpr (LN == 4 && MT <= 4) {
            if (pl.u == 1) hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 1, true, false, T>), grid, block, pl.lds_bytes, st, p);
            else hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 2, true, false, T>), grid, block, pl.lds_bytes, st, p);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    switch (pl.u) {
        case 1: hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 1, false, false, T>), grid, block, pl.lds_bytes, st, p); break;
        case 2: hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 2, false, false, T>), grid, block, pl.lds_bytes, st, p); break;
        case 4: hipLaunchKernelGGL((gemv_q4_f16_mfma_kernel<LN, MT, 4, false, false, T>), grid, block, pl.lds_bytes, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// 16-column strips always; 32- / 64-column strips: fp16 plain layers at up to 4 rows (the planner's choice from 8192 / 12288 columns up)
template <int MT, typename T>
static hipError_t launch_mfma_mt(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    if (pl.ln == 4) return launch_mfma_u<4, MT, T>(pl, p, st);
    if constexpr (std::is_same_v<T, f16> && MT <= 4) {
        if (pl.use_seq || pl.pair) return hipErrorInvalidValue;
        if (pl.ln == 8) return launch_mfma_u<8, MT, T>(pl, p, st);
        if (pl.ln == 16) return launch_mfma_u<16, MT, T>(pl, p, st);
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_mfma_t(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (pl.mt) {
        case 1: return launch_mfma_mt<1, T>(pl, p, st);
        case 2: return launch_mfma_mt<2, T>(pl, p, st);
        case 4: return launch_mfma_mt<4, T>(pl, p, st);
        case 8: return launch_mfma_mt<8, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

template <int BITS, typename T, int MT>
static hipError_t launch_mfmag_u(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    dim3 grid(pl.strips, pl.ksplit, pl.mtiles), block(pl.waves * 64);
    if (pl.ln != 4) return hipErrorInvalidValue;
    if constexpr ((BITS == 2 || BITS == 3 || BITS == 8) && !(BITS == 2 && MT == 8) && !(std::is_same_v<T, bf16> && MT == 8)) {
        if (pl.magic) {
            switch (pl.u) {
                case 1: hipLaunchKernelGGL((gemv_mfma_generic_kernel<BITS, T, 4, MT, 1, true>), grid, block, pl.lds_bytes, st, p); break;
                case 2: if constexpr (BITS == 8 || BITS == 2) { hipLaunchKernelGGL((gemv_mfma_generic_kernel<BITS, T, 4, MT, 2, true>), grid, block, pl.lds_bytes, st, p); break; } return hipErrorInvalidValue;
                case 4: if constexpr (BITS == 8) { hipLaunchKernelGGL((gemv_mfma_generic_kernel<BITS, T, 4, MT, 4, true>), grid, block, pl.lds_bytes, st, p); break; } return hipErrorInvalidValue;
                default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    if constexpr (MT <= 4) {
        switch (pl.u) {
            case 1: hipLaunchKernelGGL((gemv_mfma_generic_kernel<BITS, T, 4, MT, 1>), grid, block, pl.lds_bytes, st, p); break;
            case 2: if constexpr (BITS != 3) { hipLaunchKernelGGL((gemv_mfma_generic_kernel<BITS, T, 4, MT, 2>), grid, block, pl.lds_bytes, st, p); break; } return hipErrorInvalidValue;
            case 4: if constexpr (BITS == 4 || BITS == 8) { hipLaunchKernelGGL((gemv_mfma_generic_kernel<BITS, T, 4, MT, 4>), grid, block, pl.lds_bytes, st, p); break; } return hipErrorInvalidValue;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;                       // 8 rows per pass exist for the magic-number variants only
    }
}

template <int BITS, typename T>
static hipError_t launch_mfmag_mt(const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (pl.mt) {
        case 1: return launch_mfmag_u<BITS, T, 1>(pl, p, st);
        case 2: return launch_mfmag_u<BITS, T, 2>(pl, p, st);
        case 4: return launch_mfmag_u<BITS, T, 4>(pl, p, st);
        case 8:
            if constexpr (std::is_same_v<T, f16> && (BITS == 3 || BITS == 8)) {
                if (pl.magic && pl.waves <= 8) return launch_mfmag_u<BITS, T, 8>(pl, p, st);
            }
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

template <typename T>
static hipError_t launch_mfmag(const gptq_layer_t& L, const GemvPlan& pl, const GemvParams& p, hipStream_t st) {
    switch (L.bits) {
        case 2: return launch_mfmag_mt<2, T>(pl, p, st);
        case 3: return launch_mfmag_mt<3, T>(pl, p, st);
        case 4: if constexpr (std::is_same_v<T, bf16>) return launch_mfmag_mt<4, T>(pl, p, st); else return hipErrorInvalidValue;
        case 8: return launch_mfmag_mt<8, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

// ---- streamed kernel: plan + launch ------------------------------------------------------------------------------------
static bool stream_layer_ok(const gptq_layer_t& L) {
    if (L.g_idx != nullptr || L.epilogue != GPTQ_EPI_NONE) return false;
    if (L.bits == 4) {
        const int gu = L.group_size / 8;
        return (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16) && L.group_size % 8 == 0 && gu >= 2 && (gu & (gu - 1)) == 0 && L.K % 8 == 0;
    }
    if ((L.bits == 2 || L.bits == 3 || L.bits == 8) && L.dtype == GPTQ_F16) {       // gemv_qx_stream_kernel: packed magic-number decode, fp16 only
        const int kpu = unit_vals(L.bits);
        const int gu = L.group_size / kpu;
        return L.group_size % kpu == 0 && gu >= 1 && (gu & (gu - 1)) == 0 && L.K % kpu == 0 && L.N % 32 == 0;
    }
    return false;
}

// Measured preferences (tools/stream_sweep.py on MI355X, rotating HBM-cold weights inside a hipGraph, M = 1, us per launch;
// gpurun_out/r2d/sweep.log, summarised in DESIGN.md section 4.1):
//   4096 x 4096    register kernel 5.02 | streamed 16-col 5.10, 32-col 5.57 (64 WGs of 64 columns cannot fill the chip, and the in-launch K split costs ~1.5-2 us)
//   4096 x 11008   register 9.64 | streamed 64-col strips, 16 waves x 8 rows, one pass: 8.22 (172 workgroups of 256-byte row segments)
//   11008 x 4096   register 9.70 | streamed 9.42 at best -> not worth a second code path
//   q|k|v   (3 x 4096 x 4096 in one launch)    8.81 with 64-col strips (192 WGs) | 15.0 as three launches
//   gate|up (2 x 4096 x 11008 in one launch)  13.47 with 32-col strips (688 WGs) | 19.0 as two launches
// Strip width follows s16 = number of 64-column strips of the whole launch: 128..256 -> 64 columns, 257..512 -> 32 columns
// (344 wide workgroups leave the second round of a 256-CU chip a third full), more -> 64 columns again.
static int stream_strips64(const gptq_layer_t* const* Ls, int n) {
    int s16 = 0;
    for (int i = 0; i < n; ++i) s16 += (Ls[i]->N + 63) / 64;
    return s16;
}
static int stream_default_ln(const gptq_layer_t* const* Ls, int n) {
    const int s16 = stream_strips64(Ls, n);
    if (s16 >= 128 && (s16 <= 256 || s16 > 512)) return 16;
    if (s16 > 256) return 8;
    return (s16 * 2 >= 128) ? 8 : 4;
}
// Layers of the larger models (K and N >= 5120: Llama-13B / 33B / 70B projections): the register kernel's geometry was tuned on the
// Llama-7B shapes and loses 15-35 % there to the streamed kernel with 32-column strips, 8 waves x 4 rows per lane (64-column strips
// from 16384 columns), at every M <= 4 and for fp16 and bf16 (tools/stream_sweep.py --shapes ..., profiles/r02_stream_sweep_large_shapes.log;
// us per launch, register -> streamed, M = 1 / 4: 5120x5120 9.6 / 10.9 -> 7.4 / 9.4; 13824x5120 19.1 / 22.6 -> 14.0 / 16.3;
// 8192x8192 12.3 / 15.4 -> 10.3 / 11.7; 5120x13824 11.5 / 17.2 -> 11.2 / 13.4; 28672x8192 35.8 / 47.3 -> 25.8 / 30.1 = 4.7 TB/s;
// 17920x6656 25.1 / 29.7 -> 17.4 / 20.6).  Smaller or flatter layers (8192x3584, 3584x8192, 8192x1024, 4096x4096): equal or slower.
static bool stream_big_single(const gptq_layer_t& L) { return L.K >= 5120 && L.N >= 5120; }

// 3- / 8-bit fp16 single layers: the streamed kernel instead of gemv_mfma_generic_kernel?  (filled in from tools/stream_sweep.py --bits)
static bool stream_preferred_qx(const gptq_layer_t& L, int M) {
    // int8 from ~40 M weights (4096x11008: 17.8 -> 14.0 us, 11008x4096: 16.3 -> 13.9; 4096x4096 stays at 7.7 on the register kernel);
    // int3 single layers stay on the register kernel (9.7 against 10.0 us): its multi-layer launches are what the streamed form is for
    (void)M;
    // int2 the same (4096x11008 8.7 -> 8.0 us, 11008x4096 10.2 -> 8.1; 4096x4096 5.0 = equal: register kernel stays)
    return (L.bits == 8 || L.bits == 2) && (size_t)L.K * L.N >= ((size_t)40 << 20);
}

bool stream_preferred(const gptq_layer_t& L, int M) {
    if (L.bits != 4) return stream_preferred_qx(L, M);
    const gptq_layer_t* one[1] = {&L};
    const int s16 = stream_strips64(one, 1);
    if (stream_big_single(L)) return true;                       // M <= 4 (plan_stream)
    // only where it was measured against the register kernel: wider than 8192 columns (128 strips of 64 fill too little of the chip)
    // and narrower than 12288 (from there on the register kernel itself runs 64-column strips)
    return s16 > 128 && s16 < 192 && L.K <= 8192 && M <= 2;
}
bool multi_preferred(const gptq_layer_t* const* layers, int n, int M) {
    // one launch for a group pays while the fixed cost of a launch matters: measured (tools/multi_shapes.py, M = 1, us, one launch vs
    // separate): 7B q|k|v 8.9 vs 15.1, 13B q|k|v 14.6 vs 22.5, 13B gate|up (71 MB) 20.7 vs 25.6, 70B GQA q|k|v 14.6 vs 22.4,
    // 70B gate|up (244 MB) 57.7 vs 55.0 -- beyond ~128 MB the layers run one by one (each then takes its own best plan)
    size_t bytes = 0;
    for (int i = 0; i < n; ++i) bytes += (size_t)layers[i]->K * layers[i]->N * layers[i]->bits / 8;
    return n >= 2 && bytes <= ((size_t)128 << 20);
}

StreamPlan plan_stream(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* tune) {
    StreamPlan pl{};
    if (n < 1 || n > 4 || M < 1 || M > 4) return pl;
    const gptq_layer_t& A = *Ls[0];
    for (int i = 0; i < n; ++i) {
        const gptq_layer_t& L = *Ls[i];
        if (!stream_layer_ok(L)) return pl;
        if (L.K != A.K || L.group_size != A.group_size || L.dtype != A.dtype || L.zero_mode != A.zero_mode || L.bits != A.bits) return pl;
    }
    pl.nseg = n;
    pl.mt = M >= 3 ? 4 : M;
    const int kpu = unit_vals(A.bits), uw = unit_words(A.bits);
    const bool q4 = A.bits == 4;
    pl.units_total = A.K / kpu;
    const int gu = A.group_size / kpu;
    const bool big1 = q4 && n == 1 && stream_big_single(A);
    // 3- / 8-bit (tools/stream_sweep.py --bits 8 / 3 --gs 32, profiles/r03_stream_sweep_int{8,3}_g32.log): 32-column strips (128-byte row segments)
    // 2-bit single layers (profiles/r03_stream_sweep_int2_g64.log): 32-column strips x 8 waves x 4 units from 8192 columns (4096x11008: 8.0 us), 16-column
    // strips x 8 waves x 2 units below (11008x4096: 8.1), no K split
    const bool q2_single = A.bits == 2 && n == 1;
    int ln = (tune && tune->lanes_n) ? tune->lanes_n : (q2_single ? (A.N >= 8192 ? 8 : 4) : (!q4 ? 8 : (big1 ? (A.N >= 16384 ? 16 : 8) : stream_default_ln(Ls, n))));
    if (ln != 4 && ln != 8 && ln != 16) return pl;
    if (!q4 && ln == 16) return pl;                               // 3- / 8-bit: 16- and 32-column strips
    const int ct = ln * 4, wr = 64 / ln;
    int strips = 0, nsum = 0;
    for (int i = 0; i < n; ++i) {
        strips += (Ls[i]->N + ct - 1) / ct;
        nsum += Ls[i]->N;
    }
    pl.ln = ln;
    pl.strips_total = strips;
    pl.nsum = nsum;
    int ks = (tune && tune->ksplit) ? tune->ksplit : 0;
    if (!ks && q2_single) ks = 1;
    if (!ks && !q4 && A.bits != 3 && n == 1) {
        // int8 single layers run best as ~512-700 small workgroups: 4096x11008 (344 strips) x 2 slices 14.0 us, 11008x4096 (128 strips) x 4 slices
        // 13.9 us, against 17.8 / 16.3 for the register kernel
        ks = strips >= 256 ? 2 : 1;
        while (strips * ks < 512 && ks < 8 && pl.units_total / (ks * 2) >= wr * 4) ks *= 2;
    }
    if (!ks) {
        ks = 1;
        while (q4 && strips * ks < 128 && pl.units_total / (ks * 2) >= wr * 4) ks *= 2;   // the in-launch combine costs ~1.5-2 us: 172 unsplit workgroups beat 344 split ones
    }
    if (ks > pl.units_total) ks = pl.units_total;
    int ups = (pl.units_total + ks - 1) / ks;
    if ((size_t)strips * 4 > WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES - WS_HEADER_EPOCH_OFFSET) return pl;   // one epoch word per strip
    // (waves, U): the smallest capacity that holds a slice in ONE pass -- everything in flight from the first cycle;
    // U rows of a lane lie in one group (U <= rows per group)
    const int ucap = gu < 8 ? gu : 8;
    int waves = 0, u = 0;
    if (tune && tune->waves && tune->reserved[0]) {
        waves = tune->waves; u = tune->reserved[0];
    } else if (!q4) {              // 3- / 8-bit: small workgroups, several per CU (int8: 4 waves x 2..4 units = 8-16 KiB in flight each; int3: 8 waves x 1 unit = 24 KiB)
        if (q2_single) {
            waves = 8;
            u = ln == 8 ? 4 : 2;
            if (u > ucap) u = ucap;
            while (u > 2 && pl.units_total % u) u /= 2;
        } else if (A.bits != 3) {  // 8-bit, 2-bit groups
            waves = 4;
            u = (strips >= 256 && (n == 1 || strips >= 512)) ? 2 : 4;
            if (u > ucap) u = ucap;
            while (u > 2 && pl.units_total % u) u /= 2;
        } else {
            waves = 8; u = 1;
        }
        while (waves > 1 && (waves / 2) * wr * u >= ups) waves /= 2;
    } else if (big1) {
        waves = 8; u = ucap < 4 ? ucap : 4;
    } else {
        if (ln == 16) {            // one pass of 16 waves x 8 rows where the slice allows it (measured best for 64-column strips)
            static const int cand[][2] = {{4, 2}, {8, 2}, {8, 4}, {16, 4}, {16, 8}};
            for (auto& c : cand) {
                if (c[1] > ucap) continue;
                waves = c[0]; u = c[1];
                if (c[0] * wr * c[1] >= ups) break;
            }
        } else {                   // narrower strips: small workgroups, several per CU, a few passes each
            waves = 8; u = 2;
            // 32-column strips of a multi-layer launch (gate|up: 688 workgroups): 4 waves x 4 rows -- the same 16 KiB per workgroup, half the
            // waves to start and to join per workgroup (round-3 sweep, profiles/r03_stream_sweep_one_hop_combine.log: 13.48 against 14.07 us)
            if (ln == 8 && n >= 2 && ucap >= 4 && pl.units_total % 4 == 0) { waves = 4; u = 4; }
            while (waves > 1 && (waves / 2) * wr * u >= ups) waves /= 2;
        }
    }
    if (q4 ? (u != 2 && u != 4 && u != 8) : (A.bits != 3 ? (u != 2 && u != 4) : (u != 1 && u != 2))) return pl;      // (2- / 8-bit: the 8-unit forms were lab-only and spilled)
    if (waves < 1 || waves > 16 || u > ucap || pl.units_total % u) return pl;
    ups = (ups + u - 1) / u * u;                        // a lane's U rows start on a multiple of U: slices do too
    pl.units_per_split = ups;
    pl.ksplit = (pl.units_total + ups - 1) / ups;       // no empty slices
    pl.waves = waves;
    pl.u = u;
    pl.lds_bytes = (size_t)waves * u * uw * 1024 + (size_t)waves * (pl.mt * ct + 4) * sizeof(float) + 16;
    if (pl.lds_bytes > 160 * 1024) return pl;
    if (pl.ksplit > 8) return pl;                          // the owner's poll is unrolled over at most 7 other slices
    pl.partial_bytes = pl.ksplit > 1 ? (size_t)(pl.ksplit - 1) * M * nsum * 8 : 0;      // {fp32, tag} granules of slices 1 ..
    pl.ok = true;
    return pl;
}

// minimum waves per SIMD the kernel is compiled for (= VGPR cap 64 / 80 / 128): no spills at these pairings
template <int LN, int MT, int U, typename T>
static constexpr int stream_wps() { return (U <= 2 && MT == 1) ? 8 : ((U <= 2 || (U == 4 && MT <= 2)) ? 6 : 4); }

template <int LN, int MT, int U, typename T>
static hipError_t launch_stream_one(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    hipLaunchKernelGGL((gemv_q4_stream_kernel<LN, MT, U, T, stream_wps<LN, MT, U, T>()>), dim3(pl.strips_total * pl.ksplit), dim3(pl.waves * 64),
                       pl.lds_bytes, st, p);
    return hipGetLastError();
}
template <int LN, int MT, typename T>
static hipError_t launch_stream_u(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    switch (pl.u) {
        case 2: return launch_stream_one<LN, MT, 2, T>(pl, p, st);
        case 4: return launch_stream_one<LN, MT, 4, T>(pl, p, st);
        case 8: return launch_stream_one<LN, MT, 8, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}
template <int LN, typename T>
static hipError_t launch_stream_mt(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    switch (pl.mt) {
        case 1: return launch_stream_u<LN, 1, T>(pl, p, st);
        case 2: return launch_stream_u<LN, 2, T>(pl, p, st);
        case 4: return launch_stream_u<LN, 4, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}
template <typename T>
static hipError_t launch_stream_t(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    switch (pl.ln) {
        case 4: return launch_stream_mt<4, T>(pl, p, st);
        case 8: return launch_stream_mt<8, T>(pl, p, st);
        case 16: return launch_stream_mt<16, T>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

// 3- / 8-bit fp16 (gemv_qx_stream_kernel)
template <int BITS, int LN, int MT, int U>
static hipError_t launch_qx_one(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    hipLaunchKernelGGL((gemv_qx_stream_kernel<BITS, LN, MT, U>), dim3(pl.strips_total * pl.ksplit), dim3(pl.waves * 64), pl.lds_bytes, st, p);
    return hipGetLastError();
}
template <int BITS, int LN, int MT>
static hipError_t launch_qx_u(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    if constexpr (BITS == 8 || BITS == 2) {
        switch (pl.u) {
            case 2: return launch_qx_one<BITS, LN, MT, 2>(pl, p, st);
            case 4: return launch_qx_one<BITS, LN, MT, 4>(pl, p, st);
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (pl.u) {
            case 1: return launch_qx_one<3, LN, MT, 1>(pl, p, st);
            case 2: return launch_qx_one<3, LN, MT, 2>(pl, p, st);
            default: return hipErrorInvalidValue;
        }
    }
}
template <int BITS, int LN>
static hipError_t launch_qx_mt(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    switch (pl.mt) {
        case 1: return launch_qx_u<BITS, LN, 1>(pl, p, st);
        case 2: return launch_qx_u<BITS, LN, 2>(pl, p, st);
        case 4: return launch_qx_u<BITS, LN, 4>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}
template <int BITS>
static hipError_t launch_qx_ln(const StreamPlan& pl, const GemvStreamParams& p, hipStream_t st) {
    switch (pl.ln) {
        case 4: return launch_qx_mt<BITS, 4>(pl, p, st);
        case 8: return launch_qx_mt<BITS, 8>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_stream(const gptq_layer_t* const* Ls, const StreamPlan& pl, const void* x, void* const* outs, int M,
                         void* ws_header, void* ws_body, hipStream_t st) {
    if (!pl.ok) return hipErrorInvalidValue;
    GemvStreamParams p{};
    const int ct = pl.ln * 4;
    int blk = 0, col = 0;
    for (int i = 0; i < pl.nseg; ++i) {
        const gptq_layer_t& L = *Ls[i];
        blk += (L.N + ct - 1) / ct;
        p.seg[i] = GemvSeg{L.qweight, L.qzeros, L.scales, L.bias, outs[i], L.N, blk, col, 0};
        col += L.N;
    }
    const gptq_layer_t& A = *Ls[0];
    p.x = x;
    p.gran = (unsigned long long*)ws_body;
    p.epochs = (unsigned*)((char*)ws_header + WS_HEADER_EPOCH_OFFSET);
    p.err = (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2;
    p.max_spins = 1u << 20;
    p.nseg = pl.nseg; p.M = M; p.K = A.K; p.zero_mode = A.zero_mode;
    p.units_total = pl.units_total; p.units_per_split = pl.units_per_split; p.ksplit = pl.ksplit;
    p.gu_shift = __builtin_ctz((unsigned)(A.group_size / unit_vals(A.bits)));
    p.nsum = pl.nsum;
    if (A.bits == 8) return launch_qx_ln<8>(pl, p, st);
    if (A.bits == 3) return launch_qx_ln<3>(pl, p, st);
    if (A.bits == 2) return launch_qx_ln<2>(pl, p, st);
    return A.dtype == GPTQ_BF16 ? launch_stream_t<bf16>(pl, p, st) : launch_stream_t<f16>(pl, p, st);
}

// Per-device, once (gptq_init): instantiations whose dynamic LDS can exceed the 64 KiB default (16 waves x U = 8).
template <int LN, int MT, typename T>
static hipError_t grant_stream() {
    return hipFuncSetAttribute((const void*)gemv_q4_stream_kernel<LN, MT, 8, T, stream_wps<LN, MT, 8, T>()>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t init_gemv_device() {
    hipError_t e = hipSuccess;
    auto acc = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    acc(grant_stream<4, 1, f16>()); acc(grant_stream<4, 2, f16>()); acc(grant_stream<4, 4, f16>());
    acc(grant_stream<16, 1, f16>()); acc(grant_stream<16, 2, f16>()); acc(grant_stream<16, 4, f16>());
    acc(grant_stream<8, 1, f16>()); acc(grant_stream<8, 2, f16>()); acc(grant_stream<8, 4, f16>());
    acc(grant_stream<8, 1, bf16>()); acc(grant_stream<8, 2, bf16>()); acc(grant_stream<8, 4, bf16>());
    acc(grant_stream<4, 1, bf16>()); acc(grant_stream<4, 2, bf16>()); acc(grant_stream<4, 4, bf16>());
    acc(grant_stream<16, 1, bf16>()); acc(grant_stream<16, 2, bf16>()); acc(grant_stream<16, 4, bf16>());
    auto grant_qx = [&](auto kern) { acc(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); };
    grant_qx(gemv_qx_stream_kernel<8, 4, 1, 4>); grant_qx(gemv_qx_stream_kernel<8, 4, 2, 4>); grant_qx(gemv_qx_stream_kernel<8, 4, 4, 4>);
    grant_qx(gemv_qx_stream_kernel<8, 8, 1, 4>); grant_qx(gemv_qx_stream_kernel<8, 8, 2, 4>); grant_qx(gemv_qx_stream_kernel<8, 8, 4, 4>);
    grant_qx(gemv_qx_stream_kernel<2, 4, 1, 4>); grant_qx(gemv_qx_stream_kernel<2, 4, 2, 4>); grant_qx(gemv_qx_stream_kernel<2, 4, 4, 4>);
    grant_qx(gemv_qx_stream_kernel<2, 8, 1, 4>); grant_qx(gemv_qx_stream_kernel<2, 8, 2, 4>); grant_qx(gemv_qx_stream_kernel<2, 8, 4, 4>);
    grant_qx(gemv_qx_stream_kernel<3, 4, 1, 2>); grant_qx(gemv_qx_stream_kernel<3, 4, 2, 2>); grant_qx(gemv_qx_stream_kernel<3, 4, 4, 2>);
    grant_qx(gemv_qx_stream_kernel<3, 8, 1, 2>); grant_qx(gemv_qx_stream_kernel<3, 8, 2, 2>); grant_qx(gemv_qx_stream_kernel<3, 8, 4, 2>);
    return e;
}

hipError_t launch_gemv(const gptq_layer_t& L, const GemvPlan& pl, const void* x, void* out, int M,
                       void* workspace, hipStream_t st) {
    GemvParams p{};
    if (pl.xperm) {                    // act-order, 2+ rows: x[:, perm] once, then the plain kernel on the re-sequenced rows
        hipError_t pe = launch_permute_columns(x, L.perm, M, L.K, L.dtype, workspace, st);
        if (pe != hipSuccess) return pe;
        x = workspace;
        workspace = (char*)workspace + pl.xperm_bytes;
    }
    p.qweight = (pl.use_seq || pl.xperm) ? L.qweight_seq : L.qweight;
    p.qzeros = L.qzeros;
    p.scales = L.scales;
    p.g_idx = pl.perk ? L.g_idx : nullptr;
    p.perm = pl.use_seq ? L.perm : nullptr;
    p.bias = L.bias;
    p.x = x;
    p.out = out;
    p.partial = (float*)workspace;
    p.M = M; p.K = L.K; p.N = L.N; p.group_size = L.group_size; p.zero_mode = L.zero_mode;
    p.pair_off = pl.pair ? L.N / 2 : 0;
    p.units_total = pl.units_total; p.units_per_split = pl.units_per_split;
    p.chunk_units = pl.chunk_units; p.ksplit = pl.ksplit;
    {
        const int gu = L.group_size / 8;
        p.gu_shift = (pl.mfma && gu > 0 && (gu & (gu - 1)) == 0) ? __builtin_ctz((unsigned)gu) : -1;
    }

    hipError_t e;
    if (pl.mfmag) {
        e = (L.dtype == GPTQ_F16) ? launch_mfmag<f16>(L, pl, p, st) : launch_mfmag<bf16>(L, pl, p, st);
    } else if (pl.mfma) {
        e = (L.dtype == GPTQ_BF16) ? launch_mfma_t<bf16>(pl, p, st) : launch_mfma_t<f16>(pl, p, st);
    } else {
        e = launch_gemv_generic(L, pl, p, st);      // gemv_generic.hip
    }
    if (e != hipSuccess) return e;
    if (pl.ksplit > 1) {
        const size_t total = (size_t)M * L.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        switch (L.dtype) {
            case GPTQ_F16:
                hipLaunchKernelGGL(gemv_reduce_kernel<f16>, dim3(blocks), dim3(256), 0, st, p.partial, (const f16*)L.bias, (f16*)out, pl.ksplit, M, L.N);
                break;
            case GPTQ_BF16:
                hipLaunchKernelGGL(gemv_reduce_kernel<bf16>, dim3(blocks), dim3(256), 0, st, p.partial, (const bf16*)L.bias, (bf16*)out, pl.ksplit, M, L.N);
                break;
            default:
                hipLaunchKernelGGL(gemv_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, p.partial, (const float*)L.bias, (float*)out, pl.ksplit, M, L.N);
        }
        e = hipGetLastError();
    }
    return e;
}

}  // namespace gptq
