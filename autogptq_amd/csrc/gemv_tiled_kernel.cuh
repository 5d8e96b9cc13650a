// gemv_tiled_kernel.cuh -- the decode-copy kernel and its launch templates, included by four translation units: gemv_tiled.hip (plain layers, XM = 0),
// gemv_tiled_act.hip (act-order layers, XM = 1), gemv_tiled_peer.hip (plain layers with the tensor-parallel epilogue, XM = 3) and gemv_tiled_pair.hip
// ([gate | up] layers with the fused SiLU * mul epilogue, XM = 4) -- separate only for build time, and so that the plain kernels carry nothing of the
// other forms.  Description: gemv_tiled.hip.
#pragma once
#include "gemv_shared.cuh"

namespace gptq {

// Kernel arguments, laid out for a short prologue: everything a workgroup needs to find its layer sits in the first bytes (loaded with the other scalars
// at kernel entry), the layer's pointers are ONE dependent scalar load.  (The first version walked GemvStreamParams::seg[] with a dependent kernarg load per
// step and divided by ksplit: ~230 instructions and five scalar-load round trips before the first weight load, ~1 us per launch against the lab kernel.)
struct TiledSeg {
    const unsigned* tq;      // qweight_tiled
    const void* cst;         // qconst_tiled
    const void* bias;
    void* out;
    int N, col0;             // columns of this layer; its first column in the concatenated partial slab
    const int* perm;         // act-order layers (ACT kernels): x position i of the copy = x[perm[i]]; else unused
};
// Tensor-parallel epilogue (gptq_forward_scatter; protocol: peer.hip): the owner workgroup of a strip stores its [M][16] outputs at the rank's column
// offset of EVERY rank's exchange buffer of this call's parity; the rank's arrival flag is raised by the collect launch behind this kernel.
struct PeerEpi {
    char* xbuf[2][GPTQ_PEER_MAX];
    unsigned* flags[GPTQ_PEER_MAX];
    unsigned* state;         // [0] gathers completed (the epoch)
    int world, rank;
    unsigned row_bytes, col_off_bytes, owners, pad_;
};
struct TiledParams {
    int blk_end[4];          // cumulative strip count up to and including layer i (unused entries: INT_MAX)
    const void* x;
    unsigned long long* gran;   // K-split exchange granules (stream_finish)
    unsigned* epochs;
    unsigned* err;
    int nseg, M, K, chunks, chunks_per_split, ksplit, gu_shift, nsum, groups, xstride, waves;
    int xraw_off;            // ACT kernels: byte offset of the raw x rows (whole K, row stride 2 K + 16) in the dynamic LDS
    unsigned max_spins;
    TiledSeg seg[4];
    PeerEpi peer;            // XM = 3 kernels only: read by the owner workgroups behind their K loop
};

// Per packing: what a lane of one chunk load holds.  The chunk is always 4 k-slots x 16 columns; a lane (k-slot, column) holds WPL consecutive words =
// KPL consecutive k of ONE column, re-encoded at load time (gptq_prepack_decode) so that the packed fp16 magic-number extraction yields the k pairs in
// the order x lies in memory:
//   4-bit  4 words = 32 k; stored nibbles k0 k2 k4 k6 k1 k3 k5 k7 per word
//   8-bit  4 words = 16 k; stored bytes   k0 k2 k1 k3 per word
//   3-bit  3 words = 32 k (one packing unit, re-encoded without straddlers): word j holds the pairs p = 5 j + i (i = 0..4) = (k 2p, k 2p + 1) at bit 3 i of
//          its low / high half; bits 15 and 31 of the three words are the bits of k30 / k31.
template <int BITS> struct TiledFmt;
template <> struct TiledFmt<4> { static constexpr int WPL = 4, KPL = 32, REC = 48, ZB = 1; };
template <> struct TiledFmt<8> { static constexpr int WPL = 4, KPL = 16, REC = 64, ZB = 2; };
template <> struct TiledFmt<3> { static constexpr int WPL = 3, KPL = 32, REC = 48, ZB = 1; };
template <> struct TiledFmt<2> { static constexpr int WPL = 2, KPL = 32, REC = 48, ZB = 1; };      // (round 6) 2 words = 32 k; word w: pair p = (k 16w + 2p, k 16w + 2p + 1) at bit 2p of its halves

template <int N_> struct WordsOf { typedef unsigned type __attribute__((ext_vector_type(N_))); };

// XM = 1 (ACT): an act-order layer -- the decode copy holds the re-sequenced rows (position i = original k perm[i], groups in sequence); the workgroup DMAs the raw
// x rows (whole K) into the LDS and gathers its slice through perm LDS -> LDS (one more barrier; + M (2 K + 16) bytes of LDS); the rest is the plain kernel.
// (XM = 2 was a GATED form -- the down projection of an MLP forming silu(g) * u while staging its input: correct, and no faster than the elementwise launch it
// removed (20.6 against 20.0 us per Llama-7B MLP: every workgroup repeats the activation); not kept.)
template <int BITS, int MT, int U, typename T, int MAXW, int XM = 0, int ZM2 = 0>
__global__ void __launch_bounds__(MAXW * 64, MAXW == 16 ? 4 : 2) gemv_tiled_kernel(TiledParams p) {
    constexpr bool BF = std::is_same_v<T, bf16>;
    constexpr bool ACT = XM == 1, PEER = XM == 3;          // 3: plain staging + the tensor-parallel epilogue (gemv_tiled_peer.hip)
    // XM = 4 (PAIR, round 5): a [gate | up] layer with the SILU_MUL epilogue (the reference's fused MLP: auto_gptq/nn_modules/fused_llama_mlp.py:131-306).  A
    // workgroup takes strip s of the gate half AND strip s of the up half (N / 32 strips further) behind ONE staged x: the first half of its waves
    // streams the one, the second half the other; SiLU(gate) * up is formed on the fp32 sums behind the cross-wave reduction (no K slices: the planner).
    constexpr bool PAIR = XM == 4;
    // XM = 5 / 6 (MULTI, round 5): FOUR / TWO adjacent strips of a layer per workgroup behind ONE staged x -- the 5..8-row form staged 64 KiB of x per
    // 16-column strip (K = 4096), which the per-CU L2 pull rate (~50 GB/s, profiles/r03_xfetch_lab.log) turned into microseconds on layers of hundreds of
    // strips; the waves split into NSTR groups, one per strip, and the cross-wave sum is per strip (no K slices: the planner).
    constexpr bool MULTI = XM == 5 || XM == 6;
    constexpr int NSTR = PAIR ? 2 : (XM == 5 ? 4 : (XM == 6 ? 2 : 1));            // strips per workgroup
    using F = TiledFmt<BITS>;
    constexpr int WPL = F::WPL, KPL = F::KPL, CKE = 4 * KPL, CHB = 64 * WPL * 4, REC = F::REC, NX = KPL / 8;      // k per chunk, bytes per chunk, x pieces per lane and chunk
    constexpr int LKPL = KPL == 32 ? 5 : 4;
    typedef typename WordsOf<WPL>::type qvec;
    unsigned m_lo, m_hi, m_b, magic;                                              // opaque constants: (q & mask) | magic is ONE v_and_or_b32 (gemv.hip)
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("s_mov_b32 %0, 0x00ff00ff" : "=s"(m_b));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    unsigned m2a, m2b, m2c, m2d, m2e;
    asm("s_mov_b32 %0, 0x00030003" : "=s"(m2a));
    asm("s_mov_b32 %0, 0x000c000c" : "=s"(m2b));
    asm("s_mov_b32 %0, 0x00300030" : "=s"(m2c));
    asm("s_mov_b32 %0, 0x00c000c0" : "=s"(m2d));
    asm("s_mov_b32 %0, 0x03000300" : "=s"(m2e));
    unsigned m3a, m3b, m3c;
    asm("s_mov_b32 %0, 0x00070007" : "=s"(m3a));
    asm("s_mov_b32 %0, 0x00380038" : "=s"(m3b));
    asm("s_mov_b32 %0, 0x01c001c0" : "=s"(m3c));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                    // scalar: the staging loops below are scalar loops
    const int col = lane & 15, kb = lane >> 4;                                    // lane = kb * 16 + col: lane-linear inside the chunk
    // every scalar argument in ONE batch of kernarg loads (the empty asm pins them here: left to itself the compiler loads them one dependent step at a time)
    int ksplit = p.ksplit, be0 = p.blk_end[0], be1 = p.blk_end[1], be2 = p.blk_end[2], nchunks = p.chunks, cps = p.chunks_per_split, K = p.K, Mrows = p.M,
        G = p.groups, xstride = p.xstride, gshift = p.gu_shift, W = p.waves;
    const char* xg = (const char*)p.x;
    asm volatile("" : "+s"(ksplit), "+s"(be0), "+s"(be1), "+s"(be2), "+s"(nchunks), "+s"(cps), "+s"(K), "+s"(Mrows), "+s"(G), "+s"(xstride), "+s"(gshift), "+s"(W), "+s"(xg));
    // workgroup -> (strip over all layers, K slice); no XCD remap: strips share nothing but x, which every L2 holds
    int sidx = blockIdx.x, ks = 0;
    if (ksplit != 1) { sidx = (int)blockIdx.x / ksplit; ks = (int)blockIdx.x - sidx * ksplit; }      // uniform branch: the division only where slices exist
    const int s = (sidx >= be0) + (sidx >= be1) + (sidx >= be2);                  // scalar compares on entry-loaded words
    const TiledSeg sg = p.seg[s];                                                 // one dependent kernarg load
    const int strip = sidx - (s == 0 ? 0 : (s == 1 ? be0 : (s == 2 ? be1 : be2)));
    const int N = sg.N;
    const int Wh = NSTR == 4 ? (W >> 2) : (NSTR == 2 ? (W >> 1) : W);             // waves per strip
    const int sel = NSTR == 1 ? 0 : ((wave >= Wh ? 1 : 0) + (NSTR == 4 ? (wave >= 2 * Wh ? 1 : 0) + (wave >= 3 * Wh ? 1 : 0) : 0));      // which of the workgroup's strips (wave-uniform); PAIR: 0 = gate, 1 = up
    const int wv = NSTR == 1 ? wave : wave - sel * Wh;
    const int strip_w = PAIR ? strip + sel * (N >> 5) : (MULTI ? strip * NSTR + sel : strip);      // the strip this WAVE streams
    const int cb = ks * cps, ce = min(cb + cps, nchunks);                         // this slice's chunks
    const int kbeg = cb * CKE, kend = min(ce * CKE, K);                           // ... and its k range: what is staged of x
    // LDS: [x: MT rows of (kend - kbeg) values, row stride + 16 B][constants: G x REC bytes][cross-wave sums]
    char* const xs = smem;                                                        // row stride xstride = chunks_per_split * CKE * 2 + 16 bytes: the 4 rows of a 4-lane group hit different banks
    char* const cs = smem + (size_t)MT * xstride;
    const size_t cpad = ((size_t)G * REC + 15) & ~(size_t)15;
    float* const red = (float*)(cs + NSTR * cpad);
    const char* const cg = (const char*)sg.cst + (size_t)(MULTI ? strip * NSTR : strip) * G * REC;      // this strip's (MULTI: these strips') constants: one contiguous run
    const char* const tb = (const char*)sg.tq + (size_t)strip_w * nchunks * CHB;  // this wave's strip of weights: one contiguous run
    const unsigned t_lane = (unsigned)lane * (WPL * 4u);
    // ---- stage x and the constants by LDS DMA: no VGPRs, issued FIRST (loads return in issue order), waited for behind the first weight burst
    {
        const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)xs, cs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cs;
        const int pieces = (kend - kbeg) >> 3;                                    // 16-byte pieces per x row
        if constexpr (XM == 0 || XM == 3 || XM == 4 || MULTI) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const char* xr = xg + ((size_t)min(m, Mrows - 1) * K + kbeg) * 2;
                for (int pc0 = wave * 64; pc0 < pieces; pc0 += W * 64)            // wave-uniform trip count
                    if (pc0 + lane < pieces) lds_dma16(xr + (size_t)(pc0 + lane) * 16, xs_lds + m * xstride + pc0 * 16);      // default cache policy: every workgroup reads x
            }
        } else if constexpr (ACT) {
            // act-order: the RAW rows, whole K (a slice's positions map to any original k), by the same DMA; the gather through perm is LDS -> LDS
            const int rpieces = K >> 3;
            const unsigned xr_lds = xs_lds + (unsigned)p.xraw_off;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const char* xr = xg + (size_t)min(m, Mrows - 1) * K * 2;
                for (int pc0 = wave * 64; pc0 < rpieces; pc0 += W * 64)
                    if (pc0 + lane < rpieces) lds_dma16(xr + (size_t)(pc0 + lane) * 16, xr_lds + m * (K * 2 + 16) + pc0 * 16);
            }
        }
        const int cpieces = ((G * REC) >> 4) * (MULTI ? NSTR : 1);                // REC is a multiple of 16 (MULTI: adjacent strips' records are adjacent, cpad = G * REC)
        for (int pc0 = wave * 64; pc0 < cpieces; pc0 += W * 64)
            if (pc0 + lane < cpieces) dma16_nt(cg + (size_t)(pc0 + lane) * 16, __builtin_amdgcn_readfirstlane(cs_lds + pc0 * 16));
        if constexpr (PAIR) {                                                     // the up strip's constants behind the gate strip's
            const char* const cg2 = cg + (size_t)(N >> 5) * G * REC;
            for (int pc0 = wave * 64; pc0 < cpieces; pc0 += W * 64)
                if (pc0 + lane < cpieces) dma16_nt(cg2 + (size_t)(pc0 + lane) * 16, __builtin_amdgcn_readfirstlane(cs_lds + (unsigned)cpad + pc0 * 16));
        }
    }
    // ACT: one thread = one 16-byte piece of the staged rows: 8 consecutive positions -> 32 contiguous bytes of perm (global, requested HERE, in front of the
    // weight loads), then 8 two-byte LDS reads per row from the raw x the DMA above delivers.  (Gathering x from global instead -- 8 scattered 2-byte
    // loads per piece, by every workgroup -- cost 1.2 - 3 us per launch: profiles/r04_tiled_sweep_act_global_gather.log.)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 pa[2][2] = {};
    const int apieces = ACT ? (kend - kbeg) >> 3 : 0;
    if constexpr (ACT) {
        const int* const pg = sg.perm + kbeg;
        if (tid < apieces) { pa[0][0] = *(const i32x4*)(pg + tid * 8); pa[0][1] = *(const i32x4*)(pg + tid * 8 + 4); }
        if (tid + W * 64 < apieces) { pa[1][0] = *(const i32x4*)(pg + (tid + W * 64) * 8); pa[1][1] = *(const i32x4*)(pg + (tid + W * 64) * 8 + 4); }
    }
    auto x_gather = [&]() {                                                       // behind the barrier that makes the raw rows visible
        const int* const pg = sg.perm + kbeg;
        const char* const xraw = smem + p.xraw_off;
        for (int pc0 = tid; pc0 < apieces; pc0 += 2 * W * 64) {
            const int pcb = pc0 + W * 64;
            const bool two = pcb < apieces;
            if (pc0 != tid) {                                                     // long K only: further pieces, their perm words loaded here
                pa[0][0] = *(const i32x4*)(pg + pc0 * 8); pa[0][1] = *(const i32x4*)(pg + pc0 * 8 + 4);
                if (two) { pa[1][0] = *(const i32x4*)(pg + pcb * 8); pa[1][1] = *(const i32x4*)(pg + pcb * 8 + 4); }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned short* xr = (const unsigned short*)(xraw + (size_t)m * (K * 2 + 16));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !two) break;
                    unsigned short v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = xr[pa[h][i >> 2][i & 3]];
                    u32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (unsigned)v[2 * i] | ((unsigned)v[2 * i + 1] << 16);
                    *(u32x4*)(xs + (size_t)m * xstride + (size_t)(h ? pcb : pc0) * 16) = o;
                }
            }
        }
    };
    // bf16 layers: x is converted IN PLACE in the LDS to fp16 with one power-of-two factor per (row, run of KPL k) -- block floating point -- so that the
    // loop below is the fp16 loop: w - z is a small integer, exact in either type, and turning every fp16 pair into bf16 costs 12 VALU per packed word (the
    // round-3 kernels' 7 - 25 % bf16 gap).  A run is what one lane sums on the matrix core before its fp32 sums meet the scale, so the factor's inverse rides
    // on the scale (xe[run][row]); it puts the run's largest |x| into [2^14, 2^15): nothing overflows (bf16 reaches 3e38, fp16 65504), elements more than
    // 2^-28 below the run's largest lose bits -- below the resolution of the fp32 sum they enter.  Products stay exact (8-bit by 11-bit significands into
    // fp32), as on the bf16 matrix core.  One thread = one 16-byte piece, the NX pieces of a run in adjacent lanes; one more barrier.
    // (Tried instead, same speed at one row and slower at four: one factor per row through an LDS atomic max, and each wave converting its own chunks'
    // x inside the K loop -- profiles/r04_tiled_sweep_bf16_v*.log.)
    // The conversion is done by every workgroup for its whole K slice: at 3 - 4 rows it costs what converting the weights costs, so those keep the bf16
    // matrix core (XC false).
    // Round 5, 4-bit bf16 layers at 1 - 2 rows (XS, instead of XC): neither operand is converted.  (q >> 4 i) & 0x000f000f OR-ed with the bf16 pattern of 128 IS the bf16 pair
    // (128 + w_k, 128 + w_k+1) -- 7 VALU per packed word where the fp16 route takes 13 (+ 12 to turn the pairs into bf16 at 3 - 4 rows) -- raw x goes to the bf16
    // matrix core, and the bias is taken out of the fp32 sum of every run: sum x (128 + w) - (128 + z) sum x = sum x (w - z), with the run sums of x (xe[run][row],
    // fp32) formed ONCE per workgroup behind the staging barrier (a read-only pass: cheaper than the block-floating rewrite).  Products are exact in fp32; the
    // cancellation costs log2(143 / 8) ~ 4 of the fp32 sum's 24 bits, far below bf16's 8; a one-hot row still returns (128 + w) - (128 + z) = w - z exactly.
    // Same-session A/B against the round-4 forms (tools/lab/tiled_noxs.sh, profiles/r05_bf16_vs_f16.log; bf16 behind fp16): 1 row 6 - 13 % -> 2 - 7 %, 2 rows
    // 11 - 27 % -> 6 - 17 %; at 3 - 4 rows the per-workgroup pass over x costs more than converting the weight pairs does (13 - 48 % -> 19 - 75 %): those keep
    // the bf16 matrix core with converted weights.
    // Round 6, 4-bit bf16 layers at 3 - 4 rows, and at 2 rows in launches below 1024 workgroups (ZM): the zero-point goes to the matrix core too.  The B operand is the raw biased pair (bias + w_k, bias + w_k+1)
    // -- bias = 128 as bf16, 1024 as fp16: (q >> 4 i) & 0x000f000f | pattern, 7 VALU per packed word -- and a SECOND accumulator chain multiplies the same x
    // by the constant pair -(bias + z): sum x (bias + w) + sum x (-(bias + z)) = sum x (w - z).  No pass over x per workgroup (XS's run sums), no block-floating
    // rewrite (XC), no conversion of decoded pairs (the 3..4-row bf16 form: 25 VALU per word).  A one-hot row returns (bias + w) - (bias + z) = w - z exactly; a
    // layer whose fields equal their zero-points gives exactly 0 (the two chains are the same instruction sequence on negated operands: their sums are exact
    // negatives); everywhere else the cancellation costs log2(bias / 8) bits of the fp32 run sums (4 for bf16, 7 for fp16: 2^-17 relative, below either type's ulp).
    // Same-session A/B (tools/session_r06_zm.sh, profiles/r06_zm_ab.log; us, bf16 product -> this form | fp16): 4 rows 4096^2 5.96 -> 5.24 | 5.2, 4096 -> 11008
    // 10.1 -> 8.4 | 8.4, 11008 -> 4096 11.1 -> 9.0 | 8.7, q|k|v 10.3 -> 8.8 | 8.7, gate|up 16.1 -> 14.2 | 12.6: bf16 19 - 30 % -> 0 - 3 % (gate|up 13 %) behind fp16.
    // NOT at 1 row, nor at 2 rows in launches of 1024+ workgroups (twice the matrix-core instructions: 11 - 17 % slower than the run-sum form on the 1376-strip gate|up launch) and
    // NOT for fp16 (its 13-VALU exact form is faster: 4096 -> 11008 7.2 -> 7.7, gate|up 12.3 -> 14.3).
#if defined(GPTQ_TILED_ZM)                                                          // lab: 1 = bf16 layers at 1..4 rows, 2 = fp16 layers too (A/B builds: tools/ab_tiled.sh)
    constexpr bool ZM = BITS == 4 && MT <= 4 && (GPTQ_TILED_ZM == 2 || (GPTQ_TILED_ZM == 1 && BF));
#else
    constexpr bool ZM = BITS == 4 && BF && (MT == 4 || (MT == 2 && ZM2 == 1));      // 2 rows: where the planner asks (launches below 1024 workgroups: 4096 -> 11008 8.9 -> 8.4 us,
                                                                                    // 11008 -> 4096 9.3 -> 8.2, q|k|v 9.3 -> 8.8; the 1376-strip gate|up launch 13.3 -> 14.8: stays on the run sums)
#endif
#ifdef GPTQ_TILED_NO_XS                                                            // lab: the round-4 bf16 forms (A/B build: tools/lab/tiled_noxs.sh)
    constexpr bool XS = false;
#else
    constexpr bool XS = BF && BITS == 4 && MT <= 2 && !ZM;
#endif
    constexpr bool XC = BF && MT <= 2 && !XS && !ZM;
    using MM = std::conditional_t<XC, f16, T>;                                    // the matrix core's operand type
    constexpr int ES = MT * 16 + 4;
    float* const xe = red + W * ES;                                               // [runs of the slice][4 rows] inverse factors (bf16 layers only; planned for)
    auto x_to_f16 = [&]() {                                                       // every thread of the workgroup, behind the staging barrier
        const int prow = (kend - kbeg) >> 3, total = prow * MT;                   // pieces per row: a multiple of NX, like the thread stride
        for (int i0 = 0; i0 < total; i0 += W * 64) {                             // uniform trip count: the shuffles below need every lane
            const int idx = i0 + tid;
            const bool ok = idx < total;
            int m = 0, pc = ok ? idx : 0;
            if constexpr (MT > 1) { m = pc / prow; pc -= m * prow; }
            u32x4* const at = (u32x4*)(xs + (size_t)m * xstride + (size_t)pc * 16);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *at;
            unsigned mx = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned a = v[i] & 0x7fff7fffu;
                mx = max(mx, max(a & 0xffffu, a >> 16));
            }
            mx = max(mx, (unsigned)__shfl_xor((int)mx, 1, 64));
            if constexpr (NX == 4) mx = max(mx, (unsigned)__shfl_xor((int)mx, 2, 64));
            const int E = (int)((mx >> 7) & 0xffu);                               // biased exponent of the run's largest |x|
            const int ms = min(max(268 - E, 1), 253);                             // 2^(ms - 127): the largest lands in [2^14, 2^15)
            const float mult = __builtin_bit_cast(float, (unsigned)ms << 23);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lo = __builtin_bit_cast(float, v[i] << 16) * mult, hi = __builtin_bit_cast(float, v[i] & 0xffff0000u) * mult;
                v[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));              // exact unless the result is an fp16 subnormal
            }
            if (ok) {
                *at = v;
                if ((pc & (NX - 1)) == 0) xe[(pc / NX) * 4 + m] = __builtin_bit_cast(float, (unsigned)(254 - ms) << 23);
            }
        }
        __syncthreads();
    };
    auto x_sums = [&]() {                                                         // XS: xe[run][row] = the fp32 sum of the run's x; one thread = one 16-byte piece, the NX pieces of a run in adjacent lanes
        const int prow = (kend - kbeg) >> 3, total = prow * MT;
        for (int i0 = 0; i0 < total; i0 += W * 64) {                             // uniform trip count: the shuffles below need every lane
            const int idx = i0 + tid;
            const bool ok = idx < total;
            int m = 0, pc = ok ? idx : 0;
            if constexpr (MT > 1) { m = pc / prow; pc -= m * prow; }
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const u32x4*)(xs + (size_t)m * xstride + (size_t)pc * 16);
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sm += __builtin_bit_cast(float, v[i] << 16) + __builtin_bit_cast(float, v[i] & 0xffff0000u);
            sm += __shfl_xor(sm, 1, 64);
            if constexpr (NX == 4) sm += __shfl_xor(sm, 2, 64);
            if (ok && (pc & (NX - 1)) == 0) xe[(pc / NX) * 4 + m] = sm;
        }
        __syncthreads();
    };
    const char* const xl = xs + (size_t)min(lane & 3, MT - 1) * xstride + kb * (KPL * 2);    // A operand: lane i of a 4-lane group carries x row i
    const char* const xl2 = xs + (size_t)min(4 + (lane & 3), MT - 1) * xstride + kb * (KPL * 2);   // MT = 8: rows 4..7, a second matrix-core step per decoded pair
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    const f16x2 k960 = {(f16)960.f, (f16)960.f}, k896 = {(f16)896.f, (f16)896.f}, k1008 = {(f16)1008.f, (f16)1008.f};
    const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f}, r8 = {(f16)0.125f, (f16)0.125f}, r64 = {(f16)0.015625f, (f16)0.015625f};
    const f16x2 k768 = {(f16)768.f, (f16)768.f}, k1020 = {(f16)1020.f, (f16)1020.f}, r4 = {(f16)0.25f, (f16)0.25f}, r256 = {(f16)0.00390625f, (f16)0.00390625f};
    auto bits_of = [&](f16x2 hv) -> unsigned {                                    // the pair as the matrix core takes it: fp16 (also bf16 layers behind x_to_f16), or
        if constexpr (BF && !XC) {                                                // fp16 -> fp32 -> bf16 (exact: small integers)
            const bf16x2 o = {(bf16)(float)hv[0], (bf16)(float)hv[1]};
            return __builtin_bit_cast(unsigned, o);
        } else {
            return __builtin_bit_cast(unsigned, hv);
        }
    };
    bool staged = false;
    for (int cbase = cb; cbase < ce; cbase += Wh * U) {
        const int c0 = cbase + wv * U;
        qvec q[U];
#pragma unroll
        for (int j = 0; j < U; ++j) q[j] = __builtin_nontemporal_load((const qvec*)(tb + ((unsigned)min(c0 + j, ce - 1) * (unsigned)CHB + t_lane)));
        if (!staged) {                                                            // first pass only (uniform): the staging DMAs are OLDER than the U loads just issued
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
            if constexpr (ACT) {
                __syncthreads();                                                  // the raw rows (every wave's DMAs) are in the LDS
                x_gather();
            }
            __syncthreads();
            if constexpr (XC) x_to_f16();
            if constexpr (XS) x_sums();
            staged = true;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int cc = min(c0 + j, ce - 1);
            const int k0 = cc * CKE + kb * KPL;                                   // first k of this lane's words
            const bool live = (c0 + j < ce) && (k0 < K);                          // a ragged last chunk: whole k-slots are missing
            const int g = min(k0 >> LKPL >> gshift, G - 1);
            const char* cp = cs + (NSTR > 1 ? (size_t)sel * cpad : (size_t)0) + g * REC;
            const unsigned short sraw = *(const unsigned short*)(cp + col * 2);
            unsigned z;
            if constexpr (F::ZB == 1) z = *(const unsigned char*)(cp + 32 + col);
            else z = *(const unsigned short*)(cp + 32 + col * 2);
            u32x4 xa[NX];
#pragma unroll
            for (int w = 0; w < NX; ++w) xa[w] = *(const u32x4*)(xl + ((unsigned)(cc - cb) * (unsigned)(CKE * 2) + w * 16u));
            u32x4 xb[MT > 4 ? NX : 1];
            if constexpr (MT > 4) {
#pragma unroll
                for (int w = 0; w < NX; ++w) xb[w] = *(const u32x4*)(xl2 + ((unsigned)(cc - cb) * (unsigned)(CKE * 2) + w * 16u));
            }
            const f16x2 c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);            // -(1024 + z)
            const qvec qv = q[j];
            f32x4 accg = {0.f, 0.f, 0.f, 0.f}, accg2 = {0.f, 0.f, 0.f, 0.f};
            auto mm = [&](int pc, int hf, unsigned b0, unsigned b1) __attribute__((always_inline)) {      // 4 k of the lane's column against x piece pc, half hf: rows 0..3 (and 4..7)
                accg = Mma4<MM>::run(u32x2{xa[pc][hf * 2], xa[pc][hf * 2 + 1]}, u32x2{b0, b1}, accg);
                if constexpr (MT > 4) accg2 = Mma4<MM>::run(u32x2{xb[pc][hf * 2], xb[pc][hf * 2 + 1]}, u32x2{b0, b1}, accg2);
            };
            if constexpr (BITS == 4 && ZM) {
                unsigned mgz;
                asm("v_mov_b32 %0, %1" : "=v"(mgz) : "n"(BF ? 0x43004300 : 0x64006400));     // bias twice: bf16 128 / fp16 1024 (ulp 1 in either)
                const unsigned nz = z * 0x00010001u + (BF ? 0xC300C300u : 0xE400E400u);     // -(bias + z) twice (z <= 16: exact)
                f32x4 accz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned qw = qv[w];
                    mm(w, 0, (qw & m_lo) | mgz, ((qw >> 4) & m_lo) | mgz);                  // (k0, k1) (k2, k3): stored nibbles 0 | 4, 1 | 5
                    mm(w, 1, ((qw >> 8) & m_lo) | mgz, ((qw >> 12) & m_lo) | mgz);          // (k4, k5) (k6, k7)
                    accz = Mma4<MM>::run(u32x2{xa[w][0], xa[w][1]}, u32x2{nz, nz}, accz);
                    accz = Mma4<MM>::run(u32x2{xa[w][2], xa[w][3]}, u32x2{nz, nz}, accz);
                }
                accg += accz;
            } else if constexpr (BITS == 4 && XS) {
                unsigned magicb;
                asm("v_mov_b32 %0, 0x43004300" : "=v"(magicb));                   // bf16 128.0 twice: its 7 mantissa bits take a 4-bit field with an ulp of 1
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned qw = qv[w];
                    mm(w, 0, (qw & m_lo) | magicb, ((qw >> 4) & m_lo) | magicb);            // (k0, k1) (k2, k3): stored nibbles 0 | 4, 1 | 5
                    mm(w, 1, ((qw >> 8) & m_lo) | magicb, ((qw >> 12) & m_lo) | magicb);    // (k4, k5) (k6, k7)
                }
            } else if constexpr (BITS == 4) {
                const f16x2 c2 = c1 + k960;                                       // -(64 + z)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned qw = qv[w], q8 = qw >> 8;
                    const f16x2 h0 = as_f16x2((qw & m_lo) | magic) + c1;          // k0,k1  (stored nibbles 0 and 4)
                    const f16x2 h1 = as_f16x2((qw & m_hi) | magic) * r16 + c2;    // k2,k3  (1 and 5)
                    const f16x2 h2 = as_f16x2((q8 & m_lo) | magic) + c1;          // k4,k5  (2 and 6)
                    const f16x2 h3 = as_f16x2((q8 & m_hi) | magic) * r16 + c2;    // k6,k7  (3 and 7)
                    mm(w, 0, bits_of(h0), bits_of(h1));
                    mm(w, 1, bits_of(h2), bits_of(h3));
                }
            } else if constexpr (BITS == 2) {
                // 16 fields per word: the fp16 mantissa takes the five pairs at bits 0..9 in place (1024 + 4^p w, times 4^-p, minus (1024 / 4^p + z): exact), the
                // three at bits 10..15 come down by a shift first -- 17 VALU per 16 k
                const f16x2 c4 = c1 + k768, c16 = c1 + k960, c64 = c1 + k1008, c256 = c1 + k1020;     // -(256 + z), -(64 + z), -(16 + z), -(4 + z)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const unsigned t = qv[w], t10 = t >> 10;
                    const unsigned p0 = bits_of(as_f16x2((t & m2a) | magic) + c1);
                    const unsigned p1 = bits_of(as_f16x2((t & m2b) | magic) * r4 + c4);
                    const unsigned p2 = bits_of(as_f16x2((t & m2c) | magic) * r16 + c16);
                    const unsigned p3 = bits_of(as_f16x2((t & m2d) | magic) * r64 + c64);
                    const unsigned p4 = bits_of(as_f16x2((t & m2e) | magic) * r256 + c256);
                    const unsigned p5 = bits_of(as_f16x2((t10 & m2a) | magic) + c1);
                    const unsigned p6 = bits_of(as_f16x2((t10 & m2b) | magic) * r4 + c4);
                    const unsigned p7 = bits_of(as_f16x2((t10 & m2c) | magic) * r16 + c16);
                    mm(2 * w, 0, p0, p1);
                    mm(2 * w, 1, p2, p3);
                    mm(2 * w + 1, 0, p4, p5);
                    mm(2 * w + 1, 1, p6, p7);
                }
            } else if constexpr (BITS == 8) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {                                     // one word = 4 k = one matrix-core step
                    const unsigned qw = qv[w], q8 = qw >> 8;
                    const f16x2 h0 = as_f16x2((qw & m_b) | magic) + c1;           // k0,k1  (stored bytes 0 and 2)
                    const f16x2 h1 = as_f16x2((q8 & m_b) | magic) + c1;           // k2,k3  (1 and 3)
                    mm(w >> 1, w & 1, bits_of(h0), bits_of(h1));
                }
            } else {
                const f16x2 c3 = c1 + k896;                                       // -(128 + z): fields at bit 3, times 1/8
                const f16x2 c6 = c1 + k1008;                                      // -(16 + z): fields at bit 6, times 1/64
                unsigned pr[16];                                                  // the 16 k pairs of the unit, in k order
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const unsigned t = qv[w], t6 = t >> 6;
                    pr[5 * w + 0] = bits_of(as_f16x2((t & m3a) | magic) + c1);
                    pr[5 * w + 1] = bits_of(as_f16x2((t & m3b) | magic) * r8 + c3);
                    pr[5 * w + 2] = bits_of(as_f16x2((t & m3c) | magic) * r64 + c6);
                    pr[5 * w + 3] = bits_of(as_f16x2((t6 & m3b) | magic) * r8 + c3);
                    pr[5 * w + 4] = bits_of(as_f16x2((t6 & m3c) | magic) * r64 + c6);
                }
                const unsigned e = ((qv[0] >> 15) & 0x00010001u) | ((qv[1] >> 14) & 0x00020002u) | ((qv[2] >> 13) & 0x00040004u);   // (k30 | k31 << 16): bits 15 / 31 of the three words
                pr[15] = bits_of(as_f16x2(e | magic) + c1);
#pragma unroll
                for (int i = 0; i < 8; ++i) mm(i >> 1, i & 1, pr[2 * i], pr[2 * i + 1]);
            }
            const float sc = DType<T>::to_f32(__builtin_bit_cast(T, sraw));
            if constexpr (XS) {
                const float* const xv = xe + ((cc - cb) * 4 + kb) * 4;           // the run's sums of x, one per row
                const float zc = (float)(128u + z);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = live ? fmaf(sc, fmaf(-zc, xv[m], accg[m]), acc[m]) : acc[m];
            } else if constexpr (XC) {
                const float* const xi = xe + ((cc - cb) * 4 + kb) * 4;           // the run's inverse block factors, one per row
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = live ? fmaf(sc * xi[m], accg[m], acc[m]) : acc[m];
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = live ? fmaf(sc, m < 4 ? accg[m] : accg2[m - 4], acc[m]) : acc[m];   // a select, not a product by 0: a dead slot's x is whatever the LDS holds
            }
        }
    }
    // ---- k-slots (two shuffles: a lane owns one column), waves (LDS), then write / publish ---------------------------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float v = acc[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[m] = v;
    }
    if (!staged) __syncthreads();                                                 // (an empty slice never took the staging barrier; the planner makes none)
    if (lane < 16) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[wave * ES + m * 16 + lane] = acc[m];
    }
    __syncthreads();
    if constexpr (PAIR) {
        // entry e = (row m, column c): gate = the sum over the first half of the waves, up = over the second half (fixed order), SiLU on the fp32 sum --
        // the arithmetic of the fused GEMV epilogue of rounds 1-3 (gemv.hip) and, up to the one rounding it saves, of silu_mul_kernel (utils.hip)
        const int NH = N >> 1;
        if (tid < MT * 16) {
            const int m = tid >> 4, c = tid & 15, n = strip * 16 + c;
            float s0 = 0.f, s1 = 0.f;
            for (int w = 0; w < Wh; ++w) { s0 += red[w * ES + tid]; s1 += red[(Wh + w) * ES + tid]; }
            if (m < Mrows && n < NH) {
                if (sg.bias) { s0 += DType<T>::to_f32(((const T*)sg.bias)[n]); s1 += DType<T>::to_f32(((const T*)sg.bias)[n + NH]); }
                const float g = s0 / (1.f + __expf(-s0));
                ((T*)sg.out)[(size_t)m * NH + n] = DType<T>::from_f32(g * s1);
            }
        }
        return;
    }
    if constexpr (MULTI) {
        // per strip: the sum over its group of waves (fixed order), bias, one rounding
        for (int e = tid; e < NSTR * MT * 16; e += W * 64) {
            const int s4 = e / (MT * 16), r = e - s4 * (MT * 16), m = r >> 4, c = r & 15, n = (strip * NSTR + s4) * 16 + c;
            float t = 0.f;
            for (int w = 0; w < Wh; ++w) t += red[(s4 * Wh + w) * ES + r];
            if (m < Mrows && n < N) {
                if (sg.bias) t += DType<T>::to_f32(((const T*)sg.bias)[n]);
                ((T*)sg.out)[(size_t)m * N + n] = DType<T>::from_f32(t);
            }
        }
        return;
    }
    T* const stage = PEER ? (T*)xs : nullptr;                                      // the staged x is dead behind the barrier above
    stream_finish<16, MT, T, TiledParams, TiledSeg>(p, sg, strip, sidx, ks, N, red, stage);
    if constexpr (PEER) if (ks == 0) {                                            // uniform: the strip's owner
        const int pworld = p.peer.world;
        __syncthreads();
        const unsigned e = __hip_atomic_load(p.peer.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;      // this gather's epoch (peer.hip)
        if (tid < pworld * MT * 2) {                                              // (peer, row, half of the strip's 32 bytes)
            const int r = tid / (MT * 2), m = (tid >> 1) % MT, q = tid & 1;
            if (m < Mrows) {
                const u32x4 v = *(const u32x4*)((const char*)stage + m * 32 + q * 16);
                char* const dst = p.peer.xbuf[e & 1u][r] + (size_t)m * p.peer.row_bytes + p.peer.col_off_bytes + (size_t)strip * 32 + q * 16;
                // a system-scope WRITE-THROUGH store (sc0 sc1): the payload goes to its home, not into this XCD's L2.  NOT a release fence per
                // workgroup: at system (and agent) scope that is an L2 write-back, and hundreds of strips doing one each cost 20 - 240 us per launch
                // (profiles/r04_tp2_same_device_fused_scatter_v1 / v2*.json).
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
        }
        // No ticket, no flag here: the rank's arrival flag is raised by its COLLECT launch, which starts behind this kernel in stream order (every store
        // above has been acknowledged by then).  A ticket drawn by every strip -- 256 .. 896 fetch-adds on one word -- measured 8 - 70 us per launch.
    }
}

// ONE compilation per (BITS, MT, U, T, XM) (round 6: there were two -- workgroups of up to 16 and of up to 8 waves -- that differed by a register or two, every
// form fits 128 VGPRs): the launch bound is the one the planner's geometry of that depth uses -- 2 chunks in flight: 16-wave workgroups (<= 320 workgroups:
// one per CU); 4 chunks in flight: 4- or 8-wave workgroups, except the four-strip form (XM = 5: 16 waves = 4 strips x 4 waves).  A forced (waves, depth)
// outside its compilation is refused by plan_tiled.
template <int U, int XM> constexpr int tiled_maxw() { return (U == 2 || XM == 5) ? 16 : 8; }
int tiled_max_waves(int u, int nstr);                                             // the same rule for the planner (gemv_tiled.hip)
template <int BITS, int MT, int U, typename T, int XM>
static hipError_t launch_tiled_one(const TiledPlan& pl, const TiledParams& p, hipStream_t st) {
    constexpr int MAXW = tiled_maxw<U, XM>();
    if (pl.waves > MAXW) return hipErrorInvalidValue;
    if constexpr (BITS == 4 && MT == 2 && std::is_same_v<T, bf16> && (XM == 0 || XM == 1)) {      // plain and act-order forms
        if (pl.zm2) {
            hipLaunchKernelGGL((gemv_tiled_kernel<BITS, MT, U, T, MAXW, XM, 1>), dim3(pl.strips_total * pl.ksplit), dim3(pl.waves * 64), pl.lds_bytes, st, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((gemv_tiled_kernel<BITS, MT, U, T, MAXW, XM>), dim3(pl.strips_total * pl.ksplit), dim3(pl.waves * 64), pl.lds_bytes, st, p);
    return hipGetLastError();
}
template <int BITS, int MT, typename T, int XM>
static hipError_t launch_tiled_u(const TiledPlan& pl, const TiledParams& p, hipStream_t st) {
    switch (pl.u) {
        case 2: if constexpr (!(XM == 6 && MT <= 2)) return launch_tiled_one<BITS, MT, 2, T, XM>(pl, p, st); else return hipErrorInvalidValue;
        case 4: return launch_tiled_one<BITS, MT, 4, T, XM>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}
template <int BITS, typename T, int XM>
static hipError_t launch_tiled_mt(const TiledPlan& pl, const TiledParams& p, hipStream_t st) {
    // multi-strip workgroups: the 3..4-row and 5..8-row forms; round 6: two strips per workgroup also at 1 - 2 rows of 4-bit layers (the planner's own geometry only:
    // 8 waves x 4 chunks) -- the 1376-strip gate|up launch 11.8 -> 11.5 us (profiles/r06_multi_low_ab.log)
    constexpr bool LOW_OK = XM != 5 && (XM != 6 || BITS == 4);
    if constexpr (!LOW_OK) {
        if (pl.mt != 4 && pl.mt != 8) return hipErrorInvalidValue;
    }
    switch (pl.mt) {
        case 1: if constexpr (LOW_OK) return launch_tiled_u<BITS, 1, T, XM>(pl, p, st); else return hipErrorInvalidValue;
        case 2: if constexpr (LOW_OK) return launch_tiled_u<BITS, 2, T, XM>(pl, p, st); else return hipErrorInvalidValue;
        case 4: return launch_tiled_u<BITS, 4, T, XM>(pl, p, st);
        case 8: if constexpr (XM != 3 && XM != 4 && BITS != 2) return launch_tiled_u<BITS, 8, T, XM>(pl, p, st); else return hipErrorInvalidValue;      // 5..8 rows: plain, act-order and multi-strip forms
        default: return hipErrorInvalidValue;
    }
}
template <typename T, int XM>
static hipError_t launch_tiled_bits(const TiledPlan& pl, const TiledParams& p, hipStream_t st) {
    switch (pl.bits) {
        case 4: return launch_tiled_mt<4, T, XM>(pl, p, st);
        case 8: return launch_tiled_mt<8, T, XM>(pl, p, st);
        case 3: return launch_tiled_mt<3, T, XM>(pl, p, st);
        case 2:                                                                   // 2 bits (round 6): plain and act-order forms, up to 4 rows
            if constexpr (XM == 0 || XM == 1) { if (pl.mt <= 4) return launch_tiled_mt<2, T, XM>(pl, p, st); }
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// grants > 64 KiB of dynamic LDS to every instantiation of one x mode (long K at 4 rows of x: the staged activations can pass the default)
template <int XM>
static hipError_t grant_tiled_lds() {
    hipError_t e = hipSuccess;
    auto grant = [&](auto kern) { hipError_t r = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (e == hipSuccess) e = r; };
    auto grant_mt = [&](auto mt) {
        constexpr int MT = decltype(mt)::value;
        auto grant_u = [&](auto bu, auto uu) {
            constexpr int B = decltype(bu)::value, U = decltype(uu)::value;
            constexpr int MAXW = tiled_maxw<U, XM>();
            grant(gemv_tiled_kernel<B, MT, U, f16, MAXW, XM>); grant(gemv_tiled_kernel<B, MT, U, bf16, MAXW, XM>);
            if constexpr (B == 4 && MT == 2 && (XM == 0 || XM == 1)) grant(gemv_tiled_kernel<B, MT, U, bf16, MAXW, XM, 1>);
        };
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
        grant_u(I4{}, I2{}); grant_u(I4{}, I4{});
        grant_u(I8{}, I2{}); grant_u(I8{}, I4{}); grant_u(I3{}, I2{}); grant_u(I3{}, I4{});
        if constexpr ((XM == 0 || XM == 1) && MT <= 4) { grant_u(I2{}, I2{}); grant_u(I2{}, I4{}); }
    };
    if constexpr (XM != 5 && XM != 6) { grant_mt(std::integral_constant<int, 1>{}); grant_mt(std::integral_constant<int, 2>{}); }
    if constexpr (XM == 6) {                                                      // the 1 - 2-row two-strip form (4 bits, 8 waves x 4 chunks)
        grant(gemv_tiled_kernel<4, 1, 4, f16, 8, 6>); grant(gemv_tiled_kernel<4, 1, 4, bf16, 8, 6>);
        grant(gemv_tiled_kernel<4, 2, 4, f16, 8, 6>); grant(gemv_tiled_kernel<4, 2, 4, bf16, 8, 6>);
    }
    grant_mt(std::integral_constant<int, 4>{});
    if constexpr (XM != 3 && XM != 4) grant_mt(std::integral_constant<int, 8>{});
    return e;
}

}  // namespace gptq
