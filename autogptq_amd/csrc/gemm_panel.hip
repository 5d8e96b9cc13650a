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
    if (L.bits != 4 || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16)) return false;
    if (L.qweight_tiled == nullptr || L.qconst_tiled == nullptr || L.tiled_cols != GPTQ_STRIP_COLS) return false;
    if (L.g_idx != nullptr && !(L.perm && L.qweight_seq)) return false;
    if (L.K % 128 || L.N % 32 || L.epilogue != GPTQ_EPI_NONE) return false;
    const int gsh = panel_gsh(L);
    if (gsh == -1 || gsh == -2) return false;                  // (32-wide groups: not instantiated)
    return M >= 64;
}

// lab: tuning.path = 3, reserved[3] = GPTQ_LAB_VARIANT_PANEL_ON, reserved[0] = 10 MT + NT (0: the planner's), reserved[1] = KP (0: the planner's)
PanelPlan plan_panel(const gptq_layer_t& L, int M, const gptq_tuning_t* tune) {
    PanelPlan pl{};
    if (!panel_ok(L, M)) return pl;
    int mt = 0, nt = 0, kp = 0;
    if (tune && tune->path == 3) {
        const int g = tune->reserved[0];
        if (g / 10 == 2 && g % 10 >= 1 && g % 10 <= 4) { mt = 2; nt = g % 10; }
        if (g / 10 == 4 && g % 10 >= 1 && g % 10 <= 2 && M >= 128) { mt = 4; nt = g % 10; }
        if (tune->reserved[1] == 4 || tune->reserved[1] == 8) kp = tune->reserved[1];
    }
    if (mt == 4) kp = 4;                                       // (128-row tiles: two 16 KiB x buffers per wave)
    if (!mt) {
        // one tile costs ~ (a + b NT) per 64-deep step (x staging + NT column blocks of dequant and MFMA); a launch is rounds of 256 workgroups
        double best = 1e30;
        for (int c = 1; c <= 4; ++c) {
            const long tiles = (long)((M + 63) / 64) * ((L.N + 32 * c - 1) / (32 * c));
            const long rounds = (tiles + 255) / 256;
            const double t = (double)rounds * (0.25 + 1.0 * c + 1.5 / (L.K / 1024.0));
            if (t < best - 1e-9) { best = t; nt = c; }
        }
        mt = 2;
    }
    if (!kp) kp = 8;
    const int steps = L.K / 64;
    if (kp > steps) kp = steps >= 4 ? 4 : 0;
    if (!kp) return pl;
    pl.mt = mt; pl.nt = nt; pl.kp = kp;
    pl.nbm = (M + 32 * mt - 1) / (32 * mt);
    pl.nbn = (L.N + 32 * nt - 1) / (32 * nt);
    pl.spw = (steps + kp - 1) / kp;
    const size_t xbytes = (size_t)kp * 2 * (32 * mt) * 128;
    const int units = 4 * mt * nt, ub = units < 128 / kp ? units : 128 / kp;
    const size_t red = (size_t)ub * kp * 1024;
    pl.lds_bytes = xbytes > red ? xbytes : red;
    pl.ok = M >= 32 * mt;
    return pl;
}

// The planner's measured preference (tools/panel_ab.py, profiles/r06_panel_ab.log).
bool panel_pays(const gptq_layer_t& L, int M) {
    static const bool lab_off = getenv("GPTQ_LAB_NO_PANEL") != nullptr;      // lab: the planner as it was before this kernel
    if (lab_off || !panel_ok(L, M)) return false;
    return false;                                              // (set from the first sweep)
}

template <typename T, int MT, int NT, int KP>
static hipError_t panel_grant_one() {
    return hipFuncSetAttribute((const void*)panel::gemm_panel_kernel<T, MT, NT, KP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T>
static hipError_t panel_grant_t() {
    hipError_t e = panel_grant_one<T, 2, 1, 4>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 2, 4>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 3, 4>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 4, 4>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 1, 8>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 2, 8>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 3, 8>();
    if (e == hipSuccess) e = panel_grant_one<T, 2, 4, 8>();
    if (e == hipSuccess) e = panel_grant_one<T, 4, 1, 4>();
    if (e == hipSuccess) e = panel_grant_one<T, 4, 2, 4>();
    return e;
}
hipError_t init_gemm_panel_device() {
    hipError_t e = panel_grant_t<f16>();
    if (e == hipSuccess) e = panel_grant_t<bf16>();
    return e;
}

template <typename T, int MT, int NT, int KP>
static void panel_launch_one(const PanelPlan& pl, const panel::PanelParams& p, hipStream_t st) {
    hipLaunchKernelGGL((panel::gemm_panel_kernel<T, MT, NT, KP, false>), dim3(pl.nbm * pl.nbn), dim3(64 * KP), pl.lds_bytes, st, p);
}
template <typename T>
static hipError_t panel_launch_t(const PanelPlan& pl, const panel::PanelParams& p, hipStream_t st) {
    const int key = pl.mt * 100 + pl.nt * 10 + pl.kp;
    switch (key) {
        case 214: panel_launch_one<T, 2, 1, 4>(pl, p, st); break;
        case 224: panel_launch_one<T, 2, 2, 4>(pl, p, st); break;
        case 234: panel_launch_one<T, 2, 3, 4>(pl, p, st); break;
        case 244: panel_launch_one<T, 2, 4, 4>(pl, p, st); break;
        case 218: panel_launch_one<T, 2, 1, 8>(pl, p, st); break;
        case 228: panel_launch_one<T, 2, 2, 8>(pl, p, st); break;
        case 238: panel_launch_one<T, 2, 3, 8>(pl, p, st); break;
        case 248: panel_launch_one<T, 2, 4, 8>(pl, p, st); break;
        case 414: panel_launch_one<T, 4, 1, 4>(pl, p, st); break;
        case 424: panel_launch_one<T, 4, 2, 4>(pl, p, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gemm_panel(const gptq_layer_t& L, const PanelPlan& pl, const void* x, void* out, int M, hipStream_t st) {
    if (!pl.ok || !panel_ok(L, M)) return hipErrorInvalidValue;
    panel::PanelParams p{};
    p.qweight = L.qweight_tiled; p.qconst = (const char*)L.qconst_tiled; p.bias = L.bias; p.x = x; p.out = out;
    p.M = M; p.K = L.K; p.N = L.N;
    p.nbm = pl.nbm; p.nbn = pl.nbn;
    p.chunks = L.K / 128;
    p.groups = (L.K + L.group_size - 1) / L.group_size;
    p.gsh = panel_gsh(L);
    p.steps = L.K / 64; p.spw = pl.spw;
    return L.dtype == GPTQ_F16 ? panel_launch_t<f16>(pl, p, st) : panel_launch_t<bf16>(pl, p, st);
}

}  // namespace gptq
