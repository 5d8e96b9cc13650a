// gemm_panel.hip -- short prompts and large batches, round 6: 129 ... ~767 rows (and 128 rows on wide layers) from the DECODE COPY as whole-K panels.
//
// Why another kernel for this band.  Up to 128 rows gemm_rows.hip (16x16x32, 16 / 32 / 64 rows per workgroup) is bound by the x its workgroups pull from the
// L2; from ~768 rows the stream-K kernel (gemm_wide_sk.hip: 128 x 256 tiles, 128 x 128 per wave) fills the chip.  Between them the launches are too small
// for that tile (M = 512 on 4096 -> 4096: 64 tiles for 256 CUs; as a stream-K partition every tile is cut in four and the publish / fix-up hop costs more
// than the K loop) and ran on the round-2 tiled kernel over the checkpoint rows at 0.18 - 0.26 of the matrix peak.  The reference serves the band with
// dequant + cublasHgemm (exllamav2/cuda/q_gemm.cu:104-181, exllama/cuda_func/q4_matmul.cu:225-260) and Marlin with its stripe partition at any M
// (marlin/marlin_cuda_kernel.cu:234-300, thread-tile switch :782).
// What: a SMALLER workgroup tile that fills 256 CUs with whole-K workgroups -- 64 rows x 32 NT columns (NT = 1 .. 4), the K range split between the KP waves
// of the workgroup (K parts that meet once through LDS), nothing exchanged between workgroups: M = 512 on 4096^2 is 8 x 32 = 256 tiles of 64 x 128,
// M = 256 is 4 x 64 = 256 tiles of 64 x 64.  The body (gemm_panel_kernel.cuh) is the stream-K kernel's step on a 64-row wave tile with wave-private x
// buffers and no barrier in the K loop.  Every row tile dequantises its columns again (M / 64 times the layer: 13 VALU per 8 weights against 2 MFMAs of
// 32 cycles), which is why the tile is not smaller and why the band ends where the 128-row wave tile fills its rounds.
#include <cstdlib>

#include "gemm_panel_kernel.cuh"
#include "launch.h"

namespace gptq {

static int panel_gsh(const gptq_layer_t& L) {                 // group of the 64-deep step kt = kt >> gsh; -2: 32-wide groups; -1: not served
    if (L.group_size == 32) return -2;
    if (L.group_size >= L.K) return 30;
    if (L.group_size % 64) return -1;
    const int q = L.group_size / 64;
    if (q & (q - 1)) return -1;
    int s = 0;
    while ((1 << s) < q) ++s;
    return s;
}

bool panel_ok(const gptq_layer_t& L, int M) {
    if ((L.bits != 4 && L.bits != 3 && L.bits != 8) || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16)) return false;
    if (L.qweight_tiled == nullptr || L.qconst_tiled == nullptr || L.tiled_cols != GPTQ_STRIP_COLS) return false;
    if (L.g_idx != nullptr && !(L.perm && L.qweight_seq)) return false;
    if (L.K % 128 || L.N % 32 || L.epilogue != GPTQ_EPI_NONE) return false;
    const int gsh = panel_gsh(L);
    if (gsh == -1) return false;
    return M >= 17;                                                // (below a full 64-row panel: the x DMAs past the last row re-read it, gemm_panel_kernel.cuh)
}

// Time model of a launch, us (fit of profiles/r06_panel_sweep_cold.log: 4-bit g128 fp16, rotating HBM-cold layers): a workgroup owns its CU (128 KiB of LDS), so a
// launch is whole rounds of 256 tiles; a tile costs ~4 us (first loads from HBM, the cross-wave sum, the stores) + its 64-deep steps per wave x (0.40 + 0.50 NT) us
// (x staging + NT column blocks of dequant and MFMA, two waves per SIMD); + ~1.5 us for the launch itself.
static double panel_model_us(const gptq_layer_t& L, int M, int nt, long* tiles_out) {
    const long tiles = (long)((M + 63) / 64) * ((L.N + 32 * nt - 1) / (32 * nt));
    const long rounds = (tiles + 255) / 256;
    const int spw = (L.K / 64 + 7) / 8;
    if (tiles_out) *tiles_out = tiles;
    // 3 bits measured like 4 (profiles/r06_panel_b38.log); 8 bits: twice the packed bytes per step (+ ~10 %); 32-wide groups: constants every step; a 32-column
    // tile is bound by its pull of x (64 rows x K per 32 columns): ~1.05 us per step whatever the packing
    const double per_nt = (L.bits == 8 ? 0.55 : 0.50) + (L.group_size == 32 ? 0.02 : 0.0);
    const double step = 0.40 + per_nt * nt;
    return 1.5 + (double)rounds * (4.0 + spw * (nt == 1 && step < 1.05 ? 1.05 : step));
}

// lab: tuning.path = 3, reserved[3] = GPTQ_LAB_VARIANT_PANEL_ON, reserved[0] = 20 + NT (column blocks of 32 per workgroup tile; 0: the planner's)
PanelPlan plan_panel(const gptq_layer_t& L, int M, const gptq_tuning_t* tune) {
    PanelPlan pl{};
    if (!panel_ok(L, M)) return pl;
    int nt = 0;
    if (tune && tune->path == 3) {
        const int g = tune->reserved[0];
        if (g / 10 == 2 && g % 10 >= 1 && g % 10 <= (L.bits == 8 ? 3 : 4)) nt = g % 10;
        if (nt == 4 && M < 64) return pl;                       // a partial single panel: tiles of up to three column blocks (the four-block forms have no register for the clamped x rows)
    }
    if (!nt) {
        double best = 1e30;
        for (int c = (L.bits == 8 || M < 64 ? 3 : 4); c >= 1; --c) {      // ties go to the wider tile (fewer pulls of x); 8 bits: three register sets of 8 words per column block leave room for 3 blocks
            const double t = panel_model_us(L, M, c, nullptr);
            if (t < best - 1e-9) { best = t; nt = c; }
        }
    }
    const int steps = L.K / 64;
    pl.mt = 2; pl.nt = nt; pl.kp = 8;
    pl.nbm = (M + 63) / 64;
    pl.nbn = (L.N + 32 * nt - 1) / (32 * nt);
    pl.spw = (steps + pl.kp - 1) / pl.kp;                        // (fewer steps than waves: the last waves run empty)
    const size_t xbytes = (size_t)pl.kp * 2 * 64 * 128;
    const int units = 8 * nt, ub = units < 128 / pl.kp ? units : 128 / pl.kp;
    const size_t red = (size_t)ub * pl.kp * 1024;
    pl.lds_bytes = xbytes > red ? xbytes : red;
    pl.ok = true;
    return pl;
}

// The planner's measured preference (tools/panel_ab.py against the planner without this kernel, rotating HBM-cold layers: profiles/r06_panel_sweep_cold.log; us per
// layer call, before -> this kernel):
//   4096^2      M = 96 / 128 / 192 / 256 / 320 / 384 / 512 / 640:  13.0 / 13.5 / 17.1 / 18.3 / 27.8 / 28.7 / 35.8 / 39.9  ->  12.3 / 12.2 / 13.9 / 16.2 / 19.3 / 21.9 / 25.7 / 38.1
//   4096x11008  M = 64 / 128 / 192 / 256 / 320:                    17.4 / 28.4 / 40.8 / 42.8 / 57.4                        ->  15.4 / 21.7 / 35.8 / 40.0 / 45.6
//   11008x4096  M = 192 / 256 / 320 / 448 / 512:                   36.0 / 40.4 / 58.1 / 65.4 / 63.3                        ->  32.3 / 34.8 / 41.8 / 51.7 / 54.5
// and 1.1 - 1.4x on 2048^2 (256+ rows), 5120^2 and 8192^2 (up to 256 rows), 5120x13824 / 13824x5120 (up to 256 rows), the 70B shards 1024x8192 and 8192x3584.
// It LOSES where its tiles leave CUs idle (fewer than ~160 tiles: 4096^2 at 64 rows 0.85x, 8192x1024 and 28672x1024 up to 256 rows 0.5 - 0.87x: the rows kernel keeps
// those), on deep layers at few rows (11008x4096 at 96 / 128 rows 0.88 / 0.98x against the rows kernel; 28672x1024 at 512 rows 0.91x against the tiled one), against the
// stream-K kernel where that is the default and the launch is several rounds of tiles (512+ rows on the large shapes: 4096x11008 0.89x, 8192^2 0.91x, 5120x13824
// 0.85x) -- except on narrow layers, where stream-K has too few 256-column tiles (2048^2, 8192x1024 at 768 rows: 2.0 - 2.5x) -- and on the largest layers.
bool panel_pays(const gptq_layer_t& L, int M) {
    static const bool lab_off = getenv("GPTQ_LAB_NO_PANEL") != nullptr;      // lab: the planner as it was before this kernel
    // (the narrowest layers -- the 70B attention shard at TP = 8: stream-K has four column tiles -- up to 1536 rows: 8192x1024 at 1536 rows 47.6 -> 37.8 us; 2048 columns or
    // K = 28672 measured equal or worse and keep 1024: profiles/r06_m_sweep_prefill.log)
    if (lab_off || !panel_ok(L, M) || M > ((L.N <= 1024 && L.K <= 8192) ? 1536 : 1024)) return false;
    const PanelPlan pp = plan_panel(L, M, nullptr);
    if (!pp.ok) return false;
    long tiles = 0;
    const double est = panel_model_us(L, M, pp.nt, &tiles);
    const long rounds = (tiles + 255) / 256;
    if ((double)tiles < 0.62 * (double)(rounds * 256)) return false;
    const size_t kn = (size_t)L.K * L.N;
    if (wide_sk_pays(L, M)) {
        if (L.N <= 2048) return true;
        const double gf = 2.0 * M * (double)kn * 1e-9;
        // stream-K on cold weights: ~20 us of first loads, segment turn-around and fix-up + the K loop at ~1.2 PFLOP/s; its 3-bit / 32-wide-group forms start earlier
        // (256+ rows) and cost ~24 us + the loop, the 8-bit form ~27 us (r06_panel_b38.log: 4096x11008 at 256 / 384 / 512 rows int3 g32 43.2 / 52.0 / 60.3, int8 48.5 / 58.5 / 68.8)
        const double fixed = L.bits == 8 ? 27.0 : ((L.bits == 3 || L.group_size == 32) ? 24.0 : 20.0);
        return est < fixed + gf / 1.2;
    }
    // ONE (partial) row panel, 17 .. 64 rows, on the WIDE layers of every family (N >= 8192, K <= 8192; tools/panel_ab.py, profiles/r06_panel_big.log, default -> panel at
    // 40 / 64 rows): 5120x13824 28.3 / 29.9 -> 18.2 / 20.5 us, 6656x17920 57.0 / 58.2 -> 26.9 / 29.0, 8192x28672 72.5 / 73.9 -> 48.2 / 42.6 (the round-2 tiled kernel
    // served those: the rows kernel stops at 64 Mi weights), 8192^2 21.1 / 21.9 -> 18.9 / 22.5; 17 .. 32 rows only where the rows kernel does not serve the layer
    // (4096x11008 at 32 rows: rows 13.5, one partial panel 13.4)
    if (M <= 64 && L.N >= 8192 && L.K <= 8192) return M >= 33 || !rows_pays(L, M);
    if (M < 33) return false;
    // deep layers at 160 .. 255 rows: also beyond 128 Mi weights / K = 16384 (tools/mid_band_sweep.py at 192 rows, 128 x 256 tiles -> this kernel: 17920x6656 75.9 -> 58.9 us,
    // 28672x8192 120.7 -> 107.9; 256 rows equal; the WIDE 8192x28672 keeps the tiled / stream-K kernels: 256 rows 150 against 118)
    const bool deep_mid = L.K > 8192 && L.N <= 8192 && M >= 160 && M < 256 && kn <= ((size_t)256 << 20) && L.K <= 32768;
    if ((kn > ((size_t)128 << 20) || L.K > 16384) && !deep_mid) return false;
    // (deep layers of 64 .. 80 Mi weights at 128 .. 159 rows: 13824x5120 at 128 rows int4 / int3 g32 / int8 g32 37.3 / 37.9 / 47.5 us against 36.4 / 45.9 / 57.0 on the rows
    // kernel and 42.3 on the 128 x 256-tile one)
    if (M < 160 && L.K > 8192 && !(M >= 128 && kn > ((size_t)64 << 20) && kn <= ((size_t)80 << 20))) return false;
    // 64 .. 95 rows: the wide layers (4096x11008: 172 tiles of 64 x 64 against the rows kernel's 230 workgroups), and -- from 65 rows, i.e. two row panels -- wherever
    // the tiles fill a round (profiles/r06_panel_65_95.log, panel against the default: 4096^2 1.07 - 1.10x, 2048x4096 1.15x, 8192^2 1.08 - 1.20x, 5120x13824 1.09 - 1.14x;
    // 160 tiles of 256: 5120^2 0.87x, 13824x5120 0.93x; half a round or less: 2048^2 0.98x, 4096x2048 0.82x, 8192x1024 0.62x -- those keep the rows kernel)
    // 33 .. 63 rows: ONE partial panel where the rows kernel needs two row tiles = two rounds of workgroups -- the wide layers only (profiles/r06_m_sweep*.log,
    // 4096x11008 at 48 against 64 rows: int4 16.9 / 15.2, act-order 19.6 / 18.5, int3 g32 20.1 / 16.0, int8 g32 23.8 / 19.8 us)
    if (M <= 96 && !(L.N >= 8192 && L.K <= 4096) && (M <= 64 || (double)tiles < 0.8 * (double)(rounds * 256))) return false;      // (96 rows included: 5120^2 17.5 against the rows kernel's 14.7)
    if (M <= 96 && L.K >= 8192 && rows_pays(L, M)) return false;          // K = 8192 where the rows kernel is the alternative: 8192x3584 at 96 rows 18.7 against 16.1 (8192^2: rows stop at 64 -- this kernel)
    return true;
}

// The planner's own choice AND tiles that fill their rounds (>= 0.8): what a group call compares its one-launch kernels with (capi.hip, separate_panels_pay)
bool panel_pays_filled(const gptq_layer_t& L, int M) {
    if (!panel_pays(L, M)) return false;
    const PanelPlan pp = plan_panel(L, M, nullptr);
    long tiles = 0;
    panel_model_us(L, M, pp.nt, &tiles);
    const long rounds = (tiles + 255) / 256;
    return (double)tiles >= 0.8 * (double)(rounds * 256);
}

template <typename T, int BITS, int NT, bool G32>
static hipError_t panel_grant_one() {
    return hipFuncSetAttribute((const void*)panel::gemm_panel_kernel<T, BITS, NT, G32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T, int BITS, bool G32>
static hipError_t panel_grant_nt() {
    hipError_t e = panel_grant_one<T, BITS, 1, G32>();
    if (e == hipSuccess) e = panel_grant_one<T, BITS, 2, G32>();
    if (e == hipSuccess) e = panel_grant_one<T, BITS, 3, G32>();
    if constexpr (BITS != 8) { if (e == hipSuccess) e = panel_grant_one<T, BITS, 4, G32>(); }
    return e;
}
template <typename T>
static hipError_t panel_grant_t() {
    hipError_t e = panel_grant_nt<T, 4, false>();
    if (e == hipSuccess) e = panel_grant_nt<T, 4, true>();
    if (e == hipSuccess) e = panel_grant_nt<T, 3, false>();
    if (e == hipSuccess) e = panel_grant_nt<T, 3, true>();
    if (e == hipSuccess) e = panel_grant_nt<T, 8, false>();
    if (e == hipSuccess) e = panel_grant_nt<T, 8, true>();
    return e;
}
hipError_t init_gemm_panel_device() {
    hipError_t e = panel_grant_t<f16>();
    if (e == hipSuccess) e = panel_grant_t<bf16>();
    return e;
}

template <typename T, int BITS, int NT, bool G32>
static void panel_launch_one(const PanelPlan& pl, const panel::PanelParams& p, hipStream_t st) {
    hipLaunchKernelGGL((panel::gemm_panel_kernel<T, BITS, NT, G32>), dim3(pl.nbm * pl.nbn), dim3(512), pl.lds_bytes, st, p);
}
template <typename T, int BITS, bool G32>
static hipError_t panel_launch_nt(const PanelPlan& pl, const panel::PanelParams& p, hipStream_t st) {
    switch (pl.nt) {
        case 1: panel_launch_one<T, BITS, 1, G32>(pl, p, st); break;
        case 2: panel_launch_one<T, BITS, 2, G32>(pl, p, st); break;
        case 3: panel_launch_one<T, BITS, 3, G32>(pl, p, st); break;
        case 4: if constexpr (BITS != 8) { panel_launch_one<T, BITS, 4, G32>(pl, p, st); break; } else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <typename T>
static hipError_t panel_launch_t(int bits, bool g32, const PanelPlan& pl, const panel::PanelParams& p, hipStream_t st) {
    if (pl.mt != 2 || pl.kp != 8) return hipErrorInvalidValue;
    if (bits == 4) return g32 ? panel_launch_nt<T, 4, true>(pl, p, st) : panel_launch_nt<T, 4, false>(pl, p, st);
    if (bits == 3) return g32 ? panel_launch_nt<T, 3, true>(pl, p, st) : panel_launch_nt<T, 3, false>(pl, p, st);
    return g32 ? panel_launch_nt<T, 8, true>(pl, p, st) : panel_launch_nt<T, 8, false>(pl, p, st);
}

hipError_t launch_gemm_panel(const gptq_layer_t& L, const PanelPlan& pl, const void* x, void* out, int M, hipStream_t st) {
    if (!pl.ok || !panel_ok(L, M)) return hipErrorInvalidValue;
    panel::PanelParams p{};
    p.qweight = L.qweight_tiled; p.qconst = (const char*)L.qconst_tiled; p.bias = L.bias; p.x = x; p.out = out;
    p.M = M; p.K = L.K; p.N = L.N;
    p.nbm = pl.nbm; p.nbn = pl.nbn;
    p.chunks = L.bits == 8 ? L.K / 64 : L.K / 128;           // the decode copy's chunks: 4 k-slots of 32 (8 bits: 16) values
    p.groups = (L.K + L.group_size - 1) / L.group_size;
    const int gsh = panel_gsh(L);
    p.gsh = gsh < 0 ? 0 : gsh;                                 // (32-wide groups: group 2 kt + lane half, in the kernel)
    p.steps = L.K / 64; p.spw = pl.spw;
    return L.dtype == GPTQ_F16 ? panel_launch_t<f16>(L.bits, gsh == -2, pl, p, st) : panel_launch_t<bf16>(L.bits, gsh == -2, pl, p, st);
}

}  // namespace gptq
