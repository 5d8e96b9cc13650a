// gemm_rows.hip -- batched decode, round 5 (late): 5 ... ~256 rows from the DECODE COPY with NO exchange between workgroups.
//
// Why another kernel for this band.  The rounds 2 - 4 kernels (gemm_mid / stream64 / strip16: checkpoint rows, K slices combined across workgroups) sit at
// 0.10 - 0.27 of the HBM roofline: every K slice pays an exchange hop (2.5 - 3 us: write-through stores, flag, poll, loads), and the lab kernel of this round
// (tools/lab/gemm_strips.hip) was sized for a per-CU L2 pull of ~50 GB/s.  tools/lab/xpull.hip measures what one workgroup per CU really pulls from a buffer
// every workgroup reads (the x of a launch): 105 - 125 GB/s per CU with 8 - 16 waves and 4 - 8 KiB in flight per wave, registers or LDS DMA alike
// (profiles/r05_xpull.log) -- close to the 64 bytes per clock of the L2 -> CU path.  At that rate a workgroup can afford to read ITS ROWS OF x OVER THE WHOLE K:
//   workgroup (pm, sg) = rows [16 RB pm, + 16 RB) x columns of S adjacent 16-column strips, the whole K; its waves split the 128-deep chunks of K between
//   them (contiguous ranges) and meet once, through LDS, at the end.  Nothing is published, nobody waits for another workgroup, the sum order is fixed.
// Per chunk a wave DMAs its rows of x (RB x 4 KiB, wave-private LDS buffers: no barrier in the K loop), loads S strip-chunks of the copy (1 KiB each, one
// lane = 32 consecutive k of one column, as the decode kernel reads them) and the strips' constants, dequantises (the exact magic-number form, 13 VALU per
// 8 weights) and feeds v_mfma_f32_16x16x32: lane (column c, k-slot g) of MFMA step w holds k = 32 g + 8 w + 0..7 of the chunk on BOTH operands -- x is read
// from LDS in that order (16-byte pieces, XOR-swizzled against the 256-byte row pitch on the GLOBAL side of the DMA).
// Tried and dropped: x straight into registers (MFMA step w = k-slot w, the weights as one dword per lane and step): no LDS at all in the loop, parity-green,
// and 1.6 - 1.8x SLOWER -- the MFMA operand layout puts the four lanes of a quad on four different rows of x, i.e. four cache lines per quad where the DMA's
// lane order keeps a quad inside one line (profiles/r05_rows_ab.log, "v2").
// Cost model (plan_rows): bytes pulled per CU = K (32 RB + 8 S) + constants; dequant VALU grows with the number of row tiles (every row tile dequantises
// its strips again); the planner picks (RB, S) per shape and row count from the measured table.
// Reference behaviour this band answers: exllamav2 serves M <= 50 from the same re-laid matrix as M = 1 (exllamav2/cuda/q_gemm.cu:118, config.h:4,
// q_gemm_kernel_gptq.cuh:39-194), cuda / cuda_old use the fused kernel below 128 rows (qlinear_cuda.py:34,212).
#include <cstdlib>

#include "gemm_rows_kernel.cuh"

namespace gptq {

// ---- host side -------------------------------------------------------------------------------------------------------------------------------
static int rows_group_mode(const gptq_layer_t& L) {
    if (L.group_size >= 128) {
        if (L.group_size >= L.K) return 0;
        const int q = L.group_size / 128;
        return (L.group_size % 128 == 0 && (q & (q - 1)) == 0) ? 0 : -1;
    }
    return L.group_size == 64 ? 1 : (L.group_size == 32 ? 2 : -1);
}

bool rows_ok(const gptq_layer_t& L, int M) {
    if ((L.bits != 4 && L.bits != 3 && L.bits != 8) || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16)) return false;
    if (L.qweight_tiled == nullptr || L.qconst_tiled == nullptr || L.tiled_cols != GPTQ_STRIP_COLS) return false;
    if (L.K % 128 || L.N % 16 || L.epilogue != GPTQ_EPI_NONE || rows_group_mode(L) < 0) return false;
    if ((size_t)M * L.K * 2 >= ((size_t)1 << 32)) return false;            // 32-bit x offsets per lane
    return M >= 1 && M <= 1024;
}

RowsPlan plan_rows(const gptq_layer_t& L, int M, const gptq_tuning_t* tune);

// The planner's measured preference (tools/rows_ab.py against the default plan, profiles/r05_rows_ab.log; us per layer call, default -> this kernel):
//   4096^2      M = 5 / 8 / 16 / 32 / 64 / 96 / 128:  7.1 / 7.3 / 7.7 / 11.1 / 11.9 / 13.2 / 13.2  ->  5.5 / 5.6 / 5.8 / 6.9 / 8.2 / 10.8 / 11.0
//   4096x11008                                        11.1 / 11.0 / 11.2 / 15.1 / 21.6 / 25.1 / 26.6  ->  8.8 / 8.8 / 8.9 / 11.3 / 14.2 / 23.4 / 25.0
//   11008x4096                                        12.6 / 12.8 / 13.2 / 15.7 / 20.3 / 28.4 / 28.9  ->  9.2 / 9.3 / 10.1 / 13.1 / 16.0 / 21.5 / 22.0
// i.e. 1.07 - 1.6x from 5 to 128 rows; 192 / 256 rows 0.82 - 1.04x (every further row tile dequantises the strips again): the older kernels keep those.
// Other model shapes (old default -> this kernel, same log, M = 8 / 16 / 32 / 64 / 128): 2048^2 1.24 - 2.26x, 4096x2048 1.25 - 2.21x, 1024x8192 and 8192x1024
// 1.09 - 2.30x, 5120^2 1.25 - 1.47x, 8192^2 1.10 - 1.18x up to 64 rows (0.87x at 128); it LOSES on the largest layers, where the launch is several rounds
// of workgroups that each pull their rows of x again: 8192x28672 0.84 - 0.87x (1.21x at 32 rows), 5120x13824 0.94 - 0.97x at 8 / 16 rows, 13824x5120
// 0.93 - 1.0x up to 32 rows, 28672x8192 0.69 - 1.03x.  Hence: layers of at most 64 Mi weights, 128 rows only up to 46 M weights and 8192 columns.
bool rows_pays(const gptq_layer_t& L, int M) {
    static const bool lab_off = getenv("GPTQ_LAB_NO_ROWS") != nullptr;      // lab (tools/session_r05_rows2.sh): the planner as it was before this kernel
    if (lab_off || !rows_ok(L, M)) return false;
    const size_t kn = (size_t)L.K * L.N;
    if (M < 5 || M > 512 || L.K < 1024 || L.N < 1024) return false;
    // DEEP layers (K > 8192: the down projections of the 13B / 30B / 70B / 8B families, which the panel kernel leaves alone below 160 rows), late round 6
    // (tools/mid_band_sweep.py, profiles/r06_mid_band_sweep.log, default -> this kernel, us): beyond 64 Mi weights 33 .. 64 rows (28672x8192 at 48 / 64 rows 77.3 / 73.7 ->
    // 56.0 / 56.8, 17920x6656 40.1 / 45.3 -> 35.3 / 36.6, 13824x5120 25.5 / 28.4 -> 24.4 / 25.5) and up to 128 rows up to 128 Mi weights (17920x6656 at 96 rows 58.4 ->
    // 45.5, 13824x5120 at 96 / 128 rows 35.9 / 42.3 -> 28.7 / 38.6, 14336x4096 31.3 / 35.2 -> 28.7 / 29.2); at 8 .. 32 rows the largest layers keep the older kernels (equal
    // or 0.7 - 0.9x: round 5's sweep), and so do the WIDE ones at any row count (8192x28672 at 48 rows: 72 against the panel kernel's 51)
    // the widest K <= 4096 layers (N >= 14336: the 8B gate / up projections, fused [gate | up] layers) at up to 32 rows (tools/mid_band_sweep.py, r06_mid_band_sweep3.log):
    // 5 .. 16 rows 12.3 - 13.6 us here against 10.7 - 11.1 on the 64-column-strip kernel, 17 .. 32 rows 15.5 - 17.0 against 13.7 - 14.8 as one partial row panel
    if (L.N >= 14336 && L.K <= 4096 && M <= 32) return false;
    // act-order layers of K >= 8192 at up to 16 rows: the permute pre-pass + this kernel 17.6 / 18.6 us (8192^2 at 8 / 16 rows) against 15.0 / 16.1 on the 64-column-strip kernel
    if (L.g_idx && L.K >= 8192 && M <= 16 && L.bits == 4) return false;
    // 3-bit layers beyond 64 Mi weights at up to 32 rows (no 64-column-strip kernel for them): 5120x13824 at 8 / 16 rows 18.0 / 18.0 -> 14.9 / 15.6 us, 13824x5120 32.7 / 19.5 -> 16.8 / 18.1
    if (L.bits == 3 && kn > ((size_t)64 << 20) && kn <= ((size_t)128 << 20) && M <= 32) return true;
    const bool deep = L.K > 8192;
    // (17920x6656 at 128 rows: 65.8 against the tiled kernel's 51.2 -- up to 96 rows there, 128 only up to 80 Mi weights)
    // the WIDE largest layers (N >= 8192) at 17 .. 32 rows: 8192x28672 at 24 / 32 rows 41.8 / 42.7 us against 46.8 / 47.7 (one partial row panel) and 45.4 - 50.6 (the older
    // kernels), 6656x17920 24.1 / 24.9 against 26.6 / 25.7; from 33 rows the panel kernel has them
    if (kn > ((size_t)64 << 20) && !deep) return L.N >= 8192 && L.K > 4096 && M >= 17 && M <= 32;
    if (kn > ((size_t)64 << 20)) return deep && M >= 33 && (M <= 64 || (M <= 96 && kn <= ((size_t)128 << 20)));      // (13824x5120 at 128 rows: the panel kernel, every packing)
    if (M <= 64) return true;
    if (deep && M <= 128 && L.N <= 8192) return true;
    if (kn > (size_t)46000000 || L.N > 8192) return false;
    // 129 .. 256 rows (short prompts): the 64-row form (4 bits: gemm_rows64_kernel, half the dequant replication) -- 4096^2 M = 160 / 192 / 256 17.0 / 18.5 / 18.8 ->
    // 14.9 - 16.6 us, 11008x4096 38.5 / 38.7 / 39.5 -> 29.4 - 34.2; 4096x11008 1.0x, 320+ rows 0.6 - 1.06x (the tiled / stream-K prefill kernels keep those)
    // (only as ONE round of workgroups: 5120^2 at 224 / 256 rows = 320 workgroups 28.5 / 28.9 -> 29.9 / 30.2; 2048^2 17.3 - 20.1 -> 7.0 - 8.4, 5120^2 up to 192 rows 24 -> 19.5)
    if (M <= 128) return true;
    if (L.bits != 4) return false;
    // Small layers (at most 8 Mi weights), where the tiled kernel has too few tiles: 1.1 - 2.2x up to 512 rows whatever the round count (2048^2 M = 320 .. 512
    // 22.2 - 23.6 -> 10.2 - 11.9 us, 4096x2048 22.3 - 24.4 -> 15.1 - 17.6, 2048x4096 24.4 - 28.2 -> 18.6 - 21.4, 8192x1024 25.3 - 26.7 -> 16.4 - 20.4, 2560^2 23.8 - 25.4 -> 13.7 - 21.5)
    if (kn <= ((size_t)8 << 20)) return true;
    const RowsPlan rp = plan_rows(L, M, nullptr);
    return rp.ok && M <= 256 && (long)rp.npm * rp.nsg <= 256;
}

// strips_of: the strip counts of the n layers of one launch (a strip group never straddles two layers); nullptr: the one layer L
static RowsPlan plan_rows_n(const gptq_layer_t& L, int M, const gptq_tuning_t* tune, const int* strips_of, int n) {
    RowsPlan pl{};
    if (!rows_ok(L, M)) return pl;
    const int one_layer = L.N / 16, chunks = L.K / 128;
    if (!strips_of) { strips_of = &one_layer; n = 1; }
    auto groups_of = [&](int cs) { long g = 0; for (int i = 0; i < n; ++i) g += (strips_of[i] + cs - 1) / cs; return g; };
    // forced geometry (lab): tuning.path = 3, reserved[0] = RB (1 / 2; 4 = the 64-row form, 4 bits), reserved[1] = S
    int rb = 0, s = 0;
    if (tune && tune->path == 3 && (tune->reserved[0] == 1 || tune->reserved[0] == 2 || (tune->reserved[0] == 4 && L.bits == 4)) && tune->reserved[1] >= 1 && tune->reserved[1] <= 6) {
        rb = tune->reserved[0];
        s = tune->reserved[1];
        if (s == 5 || (s == 6 && (rb != 2 || L.bits != 4)) || (s == 4 && rb == 1 && L.bits == 8)) s = 0;      // (forms that are not built, or spill: the asm loads must never be spilled)
    }
    if (!rb || !s) {
        // cost model, us: the workgroups of one round pull K (32 RB + 8 S) bytes each at ~110 GB/s per CU; a SIMD dequantises S strips x (K / 128) chunks x 4
        // words x ~17 issue slots (13 VALU + the MFMAs and LDS reads of its row blocks) for each of the waves it hosts
        double best = 1e30;
        for (int crb = 1; crb <= 4; crb *= 2) {
            if ((crb == 2 && M <= 16) || (crb == 4 && (M <= 128 || L.bits != 4))) continue;      // the 64-row form: no faster up to 128 rows (profiles/r05_rows_ab.log)
            for (int cs : {1, 2, 3, 4, 6}) {
                if (cs == 6 && (crb != 2 || L.bits != 4)) continue;
                if (cs == 4 && crb == 1 && L.bits == 8) continue;          // (spills at 128 registers)
                const long wgs = (long)((M + 16 * crb - 1) / (16 * crb)) * groups_of(cs);
                const long rounds = (wgs + 255) / 256;
                const double pull = (double)L.K * (32.0 * crb + 8.0 * cs) / 110e3;                       // us per workgroup
                const double valu = (double)chunks * cs * 4.0 * (17.0 + 4.0 * crb) * 4.0 / 4.0 / 2.1e3;    // us per workgroup: its waves share 4 SIMDs
                // + ~2 us per round: a workgroup owns its CU (128 KiB of LDS), so the first HBM round trip of its weights and its cross-wave sum are not hidden by the
                // next one (4096x11008 M = 16: 688 one-strip workgroups = 3 rounds 12.6 us, 230 three-strip ones 8.9)
                const double t = rounds * ((pull > valu ? pull : valu) + 2.0);
                if (t < best) { best = t; rb = crb; s = cs; }
            }
        }
    }
    pl.rb = rb; pl.s = s;
    pl.xbufs = 2;
    pl.waves = rb == 4 ? 8 : 16 / rb;                          // 128 KiB of x buffers: 2 x RB x 4 KiB per wave (64 rows: 2 x 8 KiB half chunks)
    if (pl.waves > chunks) pl.waves = chunks;
    pl.cpw = (chunks + pl.waves - 1) / pl.waves;
    pl.npm = (M + 16 * rb - 1) / (16 * rb);
    pl.nsg = (int)groups_of(s);
    pl.lds_bytes = rb == 4 ? (size_t)pl.waves * 2 * 8192 : (size_t)pl.waves * pl.xbufs * rb * 4096;
    const size_t red = (size_t)pl.waves * rb * s * 1024;        // the cross-wave sum reuses the x buffers
    if (red > pl.lds_bytes) pl.lds_bytes = red;
    pl.ok = true;
    return pl;
}

hipError_t launch_gemm_rows_b38(int bits, int dtype, int gm, const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st);      // gemm_rows_b38.hip
hipError_t init_gemm_rows_b38_device();

hipError_t init_gemm_rows_device() {
    hipError_t e = rows_grant_bits<4>();
    if (e == hipSuccess) e = init_gemm_rows_b38_device();
    return e;
}

// 1 .. 4 layers in one launch (gptq_forward_multi): the same packing, group size, K and dtype, no fused epilogue; outs[i] = layer i's [M][N_i]
bool rows_multi_ok(const gptq_layer_t* const* Ls, int n, int M) {
    if (n < 1 || n > 4) return false;
    for (int i = 0; i < n; ++i) {
        const gptq_layer_t& L = *Ls[i];
        if (!rows_ok(L, M) || L.bits != Ls[0]->bits || L.group_size != Ls[0]->group_size || L.K != Ls[0]->K || L.dtype != Ls[0]->dtype) return false;
        if ((L.g_idx != nullptr) != (Ls[0]->g_idx != nullptr) || (L.g_idx && L.perm != Ls[0]->perm)) return false;      // plain layers, or act-order layers of ONE order
    }
    return true;
}

// Several layers in one launch.  The multi-layer forms of the older kernels (gemm_stream64 up to 16 rows, gemm_mid above) have enough strips to work with and hold
// their own: same-session, gptq_forward_multi, old -> this kernel (tools/rows_multi_ab.py, profiles/r05_rows_ab.log, us): q|k|v 7B M = 8 / 16 / 32 / 64 / 128
// 11.3 / 11.4 / 15.2 / 21.7 / 35.5 -> 12.0 / 12.3 / 15.1 / 23.1 / 35.7, q|k|v 13B 13.7 / 14.0 / 19.7 / 28.2 / 45.4 -> 15.7 / 16.2 / 18.3 / 29.5 / 52.5; only the
// 1376 strips of gate|up 7B between the two older kernels' sweet spots gain: M = 32 / 64 29.0 / 41.5 -> 20.1 / 32.0 (8 / 16 / 128: 16.5 / 18.5 / 58.7 -> 16.8 / 20.1 / 57.7).
bool rows_multi_pays(const gptq_layer_t* const* Ls, int n, int M) {
    if (!rows_multi_ok(Ls, n, M)) return false;
    long strips = 0;
    for (int i = 0; i < n; ++i) {
        if (!rows_pays(*Ls[i], M)) return false;
        strips += Ls[i]->N / 16;
    }
    return M >= 24 && M <= 64 && strips >= 1024;
}

RowsPlan plan_rows(const gptq_layer_t& L, int M, const gptq_tuning_t* tune) { return plan_rows_n(L, M, tune, nullptr, 1); }

RowsPlan plan_rows_multi(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* tune) {
    if (!rows_multi_ok(Ls, n, M)) return RowsPlan{};
    int strips_of[4];
    for (int i = 0; i < n; ++i) strips_of[i] = Ls[i]->N / 16;
    return plan_rows_n(*Ls[0], M, tune, strips_of, n);
}

hipError_t launch_gemm_rows_multi(const gptq_layer_t* const* Ls, int n, const RowsPlan& pl, const void* x, void* const* outs, int M, hipStream_t st) {
    if (!pl.ok || !rows_multi_ok(Ls, n, M)) return hipErrorInvalidValue;
    const gptq_layer_t& L = *Ls[0];
    rowsk::RowsParams p{};
    int end = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < n) {
            p.seg[i].qweight = Ls[i]->qweight_tiled; p.seg[i].qconst = (const char*)Ls[i]->qconst_tiled; p.seg[i].bias = Ls[i]->bias; p.seg[i].out = outs[i];
            p.seg[i].N = Ls[i]->N; p.seg[i].strips = Ls[i]->N / 16;
            end += (Ls[i]->N / 16 + pl.s - 1) / pl.s;
            p.sg_end[i] = end;
        } else {
            p.seg[i] = p.seg[0];
            p.sg_end[i] = 0x7fffffff;
        }
    }
    if (end != pl.nsg) return hipErrorInvalidValue;
    p.x = x;
    p.M = M; p.K = L.K;
    p.chunks = L.K / 128;
    p.groups = (L.K + L.group_size - 1) / L.group_size;
    p.gshift = 31;
    if (L.group_size >= 128 && L.group_size < L.K) {
        int q = L.group_size / 128, sh = 0;
        while ((1 << sh) < q) ++sh;
        p.gshift = sh;
    }
    p.npm = pl.npm; p.nsg = pl.nsg; p.cpw = pl.cpw;
    const int gm = rows_group_mode(L);
    if (L.bits == 4) return rows_launch_bits<4>(L.dtype, gm, pl, p, st);
    return launch_gemm_rows_b38(L.bits, L.dtype, gm, pl, p, st);          // gemm_rows_b38.hip
}

hipError_t launch_gemm_rows(const gptq_layer_t& L, const RowsPlan& pl, const void* x, void* out, int M, hipStream_t st) {
    const gptq_layer_t* one[1] = {&L};
    void* outs[1] = {out};
    return launch_gemm_rows_multi(one, 1, pl, x, outs, M, st);
}

}  // namespace gptq
