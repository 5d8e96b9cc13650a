// capi.hip -- the extern "C" boundary declared in include/gptq_mi355x.h: argument validation,
// path selection, error reporting.  No allocation, no synchronisation, no global mutable state
// (the only static storage is the thread-local last-error string).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.cuh"
#include "launch.h"
#include "../../include/gptq_mi355x_lab.h"

using namespace gptq;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(GPTQ_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

bool bits_ok(int b) { return b == 2 || b == 3 || b == 4 || b == 8; }
bool dtype_ok(int d) { return d == GPTQ_F16 || d == GPTQ_BF16 || d == GPTQ_F32; }

// Shape rules of the checkpoint layout (qlinear_cuda.py:51-75): K % 32 == 0 and N % 32 == 0 are
// what `infeatures // 32 * bits` / `outfeatures // 32 * bits` silently assume.
int check_layer(const gptq_layer_t* L) {
    if (!L) return fail(GPTQ_ERR_NULL, "layer is NULL");
    if (!L->qweight || !L->qzeros || !L->scales) return fail(GPTQ_ERR_NULL, "qweight/qzeros/scales must be non-NULL");
    if (!bits_ok(L->bits)) return fail(GPTQ_ERR_UNSUPPORTED, "Only 2,3,4,8 bits are supported. (got %d)", L->bits);
    if (!dtype_ok(L->dtype)) return fail(GPTQ_ERR_UNSUPPORTED, "unsupported dtype enum %d", L->dtype);
    if (L->zero_mode != GPTQ_ZERO_WRAP && L->zero_mode != GPTQ_ZERO_NOWRAP)
        return fail(GPTQ_ERR_UNSUPPORTED, "unknown zero_mode %d", L->zero_mode);
    if (L->K <= 0 || L->N <= 0 || L->K % 32 || L->N % 32)
        return fail(GPTQ_ERR_SHAPE, "in_features (%d) and out_features (%d) must be positive multiples of 32", L->K, L->N);
    if (L->group_size <= 0) return fail(GPTQ_ERR_SHAPE, "group_size must be > 0 (resolve -1 to in_features), got %d", L->group_size);
    if ((L->qweight_seq == nullptr) != (L->perm == nullptr))
        return fail(GPTQ_ERR_NULL, "qweight_seq and perm must be given together");
    if ((L->qweight_tiled != nullptr) != (L->tiled_cols != 0) || (L->qweight_tiled != nullptr) != (L->qconst_tiled != nullptr) ||
        (L->tiled_cols != 0 && L->tiled_cols != GPTQ_STRIP_COLS))
        return fail(GPTQ_ERR_UNSUPPORTED, "qweight_tiled, qconst_tiled and tiled_cols (%d) must be given together, tiled_cols = %d", L->tiled_cols, GPTQ_STRIP_COLS);
    if (L->epilogue != GPTQ_EPI_NONE && L->epilogue != GPTQ_EPI_SILU_MUL)
        return fail(GPTQ_ERR_UNSUPPORTED, "unknown epilogue %d", L->epilogue);
    if (L->epilogue == GPTQ_EPI_SILU_MUL && L->N % 64)
        return fail(GPTQ_ERR_SHAPE, "the SILU_MUL epilogue needs out_features (%d) to be a multiple of 64 ([gate | up] halves)", L->N);
    return GPTQ_OK;
}

int check_io(const void* x, const void* out, int M) {
    if (!x || !out) return fail(GPTQ_ERR_NULL, "x/out must be non-NULL");
    if (M <= 0) return fail(GPTQ_ERR_SHAPE, "M must be > 0, got %d", M);
    return GPTQ_OK;
}

bool want_gemm(const gptq_layer_t* L, int M, const gptq_tuning_t* t) {
    if (t && t->path == 3) return true;
    if (t && (t->path == 1 || t->path == 2 || t->path == 4 || t->path == 5)) return false;
    // M = 5..8 on wide layers: the GEMV needs two matrix-core passes per weight word there, and a wide N gives the tiled kernel
    // enough 256-column tiles to fill the chip (4096x11008, M = 8: 22.2 us GEMV, 17.8 us tiled; narrower layers: GEMV wins)
    const bool wide_small_batch = M >= 5 && L->bits == 4 && L->N > 8192 && L->epilogue == GPTQ_EPI_NONE;
    if (L->dtype == GPTQ_F32) {
        // fp32 layers: the exact-f32 matrix-core kernel runs whole 128-row tiles with no K split; up to 64 rows the 4-rows-per-pass GEMV
        // is faster or equal on every shape (tools/cliff_scan.py --slice D: 4096x11008 M = 8: 517 us against ~60; M = 64: 528 against
        // ~510; 4096x4096 M = 64 on 32 tiles: 455 against 85)
        if (M <= 64) return false;
        const GemmPlan gf = plan_gemm(*L, M, t);
        return gf.supported && (long)gf.nbm * gf.nbn >= 64;      // fewer tiles than a quarter of the chip: the GEMV still wins (4096x4096 M = 128: ~170 us against 455)
    }
    if (!t && L->epilogue == GPTQ_EPI_NONE && rows_pays(*L, M)) return true;      // 5 .. 128 rows from the decode copy (gemm_rows.hip; plan_gemm picks it)
    if (L->bits != 4) {
        // 2/3/8-bit: the matrix-core GEMV handles 4 rows of x per pass over the weights and the generic (act-order) kernel fewer, so the
        // weight-streaming GEMMs take over early (tools/cliff_scan.py --slice C, 4096x11008, us: int8 M = 8: 47.6 GEMV, 25.6 tiled at
        // M = 16; int8 act-order M = 4: 43.7 against 29.5; int3 M = 8: 26.2 against 35.8 -- 3-bit keeps the GEMV up to 8 rows)
        // (act-order layers with the re-sequenced side copy run the same kernels on a permuted x -- GEMV and GEMM alike pay one 2.6 us
        // pre-pass -- so they share the plain layers' crossovers; only raw act-order layers, which sit on the fp32 generic GEMV, leave early)
        const bool raw_act = L->g_idx != nullptr && !(L->qweight_seq != nullptr && L->perm != nullptr);
        const bool big = (size_t)L->K * L->N >= ((size_t)32 << 20);
        // int8 fp16 on layers of at most 256 strips: the one-pass 8-row GEMV beats the GEMMs up to 8 rows (us at M = 8, GEMV / GEMM: 4096x4096
        // 12.5 / 16.6, 11008x4096 27.2 / 32.4; 4096x11008 26.0 / 23.6 keeps the GEMM from 5 rows)
        const bool int8_rows8 = L->bits == 8 && L->dtype == GPTQ_F16 && L->N <= 4096 && L->epilogue == GPTQ_EPI_NONE && !raw_act;
        const int min_m = raw_act ? (L->bits == 8 ? 3 : 5) : (L->bits == 8 ? (int8_rows8 ? 9 : 5) : ((L->bits == 2 && big) ? 5 : 9));
        if (M >= 5 && M < min_m && L->epilogue == GPTQ_EPI_NONE && !raw_act) {
            const GemmPlan g8 = plan_gemm(*L, M, t);              // 5..8 rows: the 2- / 3- / 8-bit forms of gemm_mid_kernel where the planner takes them (tools/nonq4_batched.py:
            if (g8.supported && g8.mid) return true;              // 11008x4096 M = 8 27.1 -> 16.1 us, 4096x4096 12.3 -> 11.1)
        }
        if (M < min_m) return false;
        return plan_gemm(*L, M, t).supported;
    }
    if (M <= 2) return false;
    const GemmPlan g = plan_gemm(*L, M, t);
    // M = 3..4: only the unsplit streamed 64-column-strip kernel (160+ strips) beats the GEMVs there (tools/cliff_scan.py, us, M = 3 / 4 / 5:
    // 4096x11008 11.8 / 12.0 / 10.8, 5120x13824 15.0 / 15.8 / 13.1 -- the M = 5 column is that kernel, whose time barely depends on M up to 16)
    if (M <= 4 && !(g.supported && g.stream64 && g.ksplit == 1 && L->epilogue == GPTQ_EPI_NONE)) return false;
    // M = 5..8: the streamed 64-column-strip kernel (one matrix-core pass for up to 16 rows, weights by LDS DMA) where it exists;
    // layers with a fused epilogue keep the GEMV that applies it
    // (and the 16-column-strip kernel on narrow layers: 4096x4096 M = 5 / 8: 7.3 / 7.9 us against the GEMV's 8.0 / 8.3)
    // (28672x1024, M = 8: GEMV 14.8, 16-column strips 17.0 -- long K keeps the GEMV)
    const bool small_batch_stream = g.supported && (g.stream64 || (g.strip16 && L->K <= 8192)) && L->epilogue == GPTQ_EPI_NONE;
    if (M <= 8 && !wide_small_batch && !small_batch_stream) return false;
    return g.supported;
}

// The streamed GEMV (weights by LDS DMA, in-launch K-split combine) for a single layer: forced by tuning.path = 6, else by
// the planner's measured preference.
bool want_stream(const gptq_layer_t* L, int M, const gptq_tuning_t* t) {
    if (M > 4 || L->epilogue != GPTQ_EPI_NONE) return false;
    if (t && t->path != 0 && t->path != 6) return false;
    const gptq_layer_t* one[1] = {L};
    const StreamPlan sp = plan_stream(one, 1, M, t);
    if (!sp.ok) return false;
    if (t && t->path == 6) return true;
    if (M >= 3) {                                      // 3..4 rows on layers of 160+ strips: the batched-decode kernel (want_gemm)
        const Stream64Plan s64 = plan_stream64(one, 1, M, nullptr);
        if (s64.ok && s64.pays && s64.ksplit == 1) return false;
    }
    return stream_preferred(*L, M);
}

// Decode from the strip-major side copy (gemv_tiled.hip): 3- / 4- / 8-bit fp16 / bf16 layers (plain or act-order) that carry qweight_tiled, M <= 4.  tuning.path = 8 forces it
// ("does not fit" is then an error), any other explicit path keeps the checkpoint-layout kernels (A/B runs).
// 5..8 rows (MT = 8: two matrix-core steps per decoded pair, 8 staged rows of x per workgroup): every 16-column strip stages its own copy of x, which is what
// this form loses to the 64-column-strip kernels on wide layers (tools/tiled_sweep.py --m 8, profiles/r04_tiled_sweep_m8.log, us, this / checkpoint-layout
// default: 11008x4096 18.7 / 12.3, 4096x11008 12.9 / 10.8, q|k|v 13.1 / 11.7, gate|up 22.1 / 18.3) and wins only on a single square-ish 4096-wide layer
// (4096^2: 6.62 / 7.20) -- the planner takes it there by itself (profiles/r04_tiled_sweep_rows8_default_rule.log, M = 8: 4096^2 6.60 / 7.29, 2048^2 4.78 / 6.89,
// 4096x2048 5.95 / 7.66, 2048x4096 5.05 / 5.25, 3072^2 6.16 / 6.10; M = 5: 6.47 / 7.11, 4.77 / 6.79), tuning.path = 8 asks for it anywhere.
constexpr int TILED_ROWS_MAX = 8;
static bool tiled_rows8_pays(const gptq_layer_t* const* Ls, int n) {
    const gptq_layer_t& L = *Ls[0];
    if (n == 1 && L.bits == 4 && !L.g_idx && L.K >= 2048 && L.K <= 4096 && L.N >= 2048 && L.N <= 4096) return true;
    // launches of 1024+ strips (gate|up of a 7B block): two strips per workgroup behind one staged x (gemv_tiled_multi.hip) -- M = 8 16.5 us against 17.3 for
    // the batched-decode kernel on the checkpoint rows (profiles/r05_multi_strip_ab.log); 8 rows of x over the whole K have to fit the LDS
    long strips = 0;
    for (int i = 0; i < n; ++i) {
        if (Ls[i]->bits != 4 || Ls[i]->g_idx || Ls[i]->N % 32) return false;
        strips += Ls[i]->N / 16;
    }
    return n >= 2 && strips >= 1024 && L.K <= 4096;
}
// plan_out: the plan the answer was derived from (the eager decode call computes it ONCE: a launch of this kernel runs for 4.5 us, the host side of an eager
// call is measured in the same unit -- tools/host_cost.py)
bool want_tiled(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* t, TiledPlan* plan_out = nullptr) {
    if (M > TILED_ROWS_MAX || n < 1 || n > 4) return false;
    if (Ls[0]->qweight_tiled == nullptr) return false;
    if (Ls[0]->epilogue != GPTQ_EPI_NONE && (n != 1 || M > 4)) return false;     // a [gate | up] layer: the pair form of the kernel (gemv_tiled_pair.hip)
    if (M > 4 && !(t && t->path == 8) && n == 1 && rows_pays(*Ls[0], M)) return false;      // 5+ rows of one layer: the exchange-free batched-decode kernel (gemm_rows.hip: 4096^2 M = 5 / 8 7.1 / 7.3 -> 5.5 / 5.6 us)
    if (M > 4 && !(t && t->path == 8) && !tiled_rows8_pays(Ls, n)) return false;
    if (t && t->path != 0 && t->path != 8) return false;
    // act-order layers at 3 - 4 rows from K = 5120: the in-kernel gather holds the raw AND the gathered rows of x in LDS (2 x 4 rows x K: one workgroup per CU from
    // K = 5120) -- the permute pre-pass + the kernels of the 5+-row path are faster there (tools/m_sweep.py --act, profiles/r06_act_m4.log, 4 rows against 5:
    // 5120x13824 26.5 / 14.7 us, 8192^2 18.3 / 14.8, 6656^2 16.7 / 15.6, 5120^2 13.1 / 12.3; 4096-deep layers keep this kernel: 4096x11008 11.7 / 12.9).  (act-order
    // layers are never released to the host: nothing assumes the decode copy for them)
    // narrow layers (tensor-parallel shards, N < 4096) keep it: 8192x3584 9.2 against 10.9, 8192x1024 8.1 / 9.8
    if (!(t && t->path == 8) && M >= 3 && M <= 4 && Ls[0]->g_idx && Ls[0]->K >= 5120 && Ls[0]->N >= 4096 && n == 1) return false;
    // ... and at 2 rows from K = 12288 (2 x 2 rows x K of LDS again leave one workgroup per CU): 13824x5120 21.4 us against 18.7 at THREE rows on the pre-pass + streamed kernel
    // (layers of more than 256 strips only: 14336x4096 runs one workgroup per CU either way and keeps this kernel)
    if (!(t && t->path == 8) && M == 2 && Ls[0]->g_idx && Ls[0]->K >= 12288 && Ls[0]->N > 4096 && n == 1) return false;
    const TiledPlan tp = plan_tiled(Ls, n, M, t);                  // act-order layers: the copy holds the re-sequenced rows, the kernel gathers x through perm
    if (plan_out) *plan_out = tp;
    return tp.ok;
}

// act-order layers on the streamed GEMV: x permuted once by the column-permute pre-pass, the kernel streams the re-sequenced rows of a
// plain copy of the layer.  Where the streamed kernel is the planner's choice for that copy (tools/cliff_scan.py --slice F, us, in-kernel
// gather / register kernel -> this: 13824x5120 M = 1 / 2 / 4 19.6 / 24.7 / 26.9 -> ~17 / 17.3 / 19.8; 8192x28672 M = 1 37.0 -> ~31); with
// one row only from 33 MiB up -- below that the in-kernel gather is cheaper than the 2.6 us pre-pass (5120x5120: 9.7 against 10.1).
static bool want_stream_seq(const gptq_layer_t* L, int M, const gptq_tuning_t* t, gptq_layer_t* P) {
    if (t || !L->g_idx || !L->qweight_seq || !L->perm || L->epilogue != GPTQ_EPI_NONE || M > 4) return false;
    *P = *L;
    P->g_idx = nullptr; P->perm = nullptr; P->qweight = L->qweight_seq; P->qweight_seq = nullptr;
    if (!want_stream(P, M, nullptr)) return false;
    if (M == 1 && (size_t)L->K * L->N * L->bits / 8 < ((size_t)33 << 20)) return false;
    return true;
}
static size_t xperm16_bytes(const gptq_layer_t* L, int M) { return ((size_t)M * L->K * 2 + 255) / 256 * 256; }

// workspace split: [ticket header | body]
struct WsView { void* header; void* body; size_t body_bytes; };
WsView split_ws(void* ws, size_t ws_bytes) {
    if (!ws || ws_bytes <= WS_HEADER_BYTES) return WsView{nullptr, nullptr, 0};
    return WsView{ws, (char*)ws + WS_HEADER_BYTES, ws_bytes - WS_HEADER_BYTES};
}

}  // namespace

extern "C" {

int gptq_abi_version(void) { return GPTQ_MI355X_ABI_VERSION; }
const char* gptq_last_error(void) { return g_err; }

const char* gptq_status_string(int s) {
    switch (s) {
        case GPTQ_OK: return "ok";
        case GPTQ_ERR_NULL: return "null pointer";
        case GPTQ_ERR_SHAPE: return "bad shape";
        case GPTQ_ERR_UNSUPPORTED: return "unsupported configuration";
        case GPTQ_ERR_WORKSPACE: return "workspace too small";
        case GPTQ_ERR_LAUNCH: return "HIP launch failure";
        default: return "unknown status";
    }
}

// Unfused SILU_MUL: y[M, N] goes to the front of the workspace, the elementwise pass writes out[M, N/2].
static size_t epi_scratch_bytes(const gptq_layer_t* L, int M) {
    const size_t b = (size_t)M * L->N * dtype_size(L->dtype);
    return (b + 255) / 256 * 256;
}
// A [gate | up] layer with 3..8 rows of x: the GEMV's fused forms for 3+ rows (MT = 4, two passes for 5..8 rows) lose to the streamed
// 64-column-strip kernel followed by the elementwise pass (4096 x 22016, tools/fused_small_batch.py: M = 4 27.7 -> 19.5 us,
// M = 5 / 8 47.6 / 50.8 -> 19.3 / 19.5 us; M = 1 / 2 stay fused: 16.7 / 21.2 us).  Returns the tuning the inner (epilogue-stripped)
// call runs with: the caller's, or -- for 3..4 rows, below the planner's own threshold for that kernel -- a local one that asks for it.
static bool small_batch_stream_for_epilogue(const gptq_layer_t* L, int M) {
    if (L->epilogue != GPTQ_EPI_SILU_MUL || M < 3 || M > 8) return false;
    gptq_layer_t Lc = *L;
    Lc.epilogue = GPTQ_EPI_NONE;
    const gptq_layer_t* one[1] = {&Lc};
    const Stream64Plan sp = plan_stream64(one, 1, M, nullptr);
    return sp.ok && sp.pays;
}
static const gptq_tuning_t* inner_tuning(const gptq_layer_t* L, int M, const gptq_tuning_t* tune, gptq_tuning_t* local) {
    if (tune) return tune;
    if (M <= 4 && small_batch_stream_for_epilogue(L, M)) {
        *local = gptq_tuning_t{};
        local->path = 3;
        local->reserved[2] = 4;
        return local;
    }
    return nullptr;
}
static bool fused_epilogue_ok(const gptq_layer_t* L, int M, const gptq_tuning_t* tune) {
    if (!tune && small_batch_stream_for_epilogue(L, M)) return false;
    return L->epilogue == GPTQ_EPI_SILU_MUL && !want_gemm(L, M, tune) && plan_gemv(*L, M, tune).pair;
}

// Body = what the kernels use behind the ticket header; the public figure adds the header whenever there is a body.
static bool tiled_pair_call(const gptq_layer_t* L, int M, const gptq_tuning_t* tune) {
    const gptq_layer_t* one[1] = {L};
    return L->epilogue == GPTQ_EPI_SILU_MUL && want_tiled(one, 1, M, tune);
}
static size_t body_bytes(const gptq_layer_t* L, int M, const gptq_tuning_t* tune) {
    if (tiled_pair_call(L, M, tune)) return 0;                                    // decode rows of a [gate | up] layer that carries its copy: one launch, no scratch
    if (L->epilogue != GPTQ_EPI_NONE && !fused_epilogue_ok(L, M, tune)) {
        gptq_layer_t Lc = *L;
        Lc.epilogue = GPTQ_EPI_NONE;
        gptq_tuning_t local;
        const size_t inner = body_bytes(&Lc, M, inner_tuning(L, M, tune, &local));          // the inner call shares the ONE zeroed ticket header; its body starts behind y
        return epi_scratch_bytes(L, M) + inner;
    }
    size_t a = plan_gemv(*L, M, tune).workspace_bytes;
    GemmPlan g = plan_gemm(*L, M, tune);
    size_t b = g.supported ? g.workspace_bytes : 0;
    if (g.supported && (g.rows || g.panel) && want_gemm(L, M, tune)) return b;                 // the exchange-free batched-decode kernel: nothing but the permuted x of act-order layers
    size_t c = 0;
    {
        const gptq_layer_t* one[1] = {L};
        if (want_tiled(one, 1, M, tune)) return plan_tiled(one, 1, M, tune).partial_bytes;
    }
    if (want_stream(L, M, tune)) {
        const gptq_layer_t* one[1] = {L};
        c = plan_stream(one, 1, M, tune).partial_bytes;
    }
    gptq_layer_t P;
    if (want_stream_seq(L, M, tune, &P)) {
        const gptq_layer_t* one[1] = {&P};
        c = std::max(c, xperm16_bytes(L, M) + plan_stream(one, 1, M, nullptr).partial_bytes);
    }
    return std::max(std::max(a, b), c);
}

size_t gptq_workspace_bytes_ex(const gptq_layer_t* L, int M, const gptq_tuning_t* tune) {
    if (check_layer(L) != GPTQ_OK || M <= 0) return 0;
    const size_t b = body_bytes(L, M, tune);
    return b ? WS_HEADER_BYTES + b : 0;
}

size_t gptq_workspace_bytes(const gptq_layer_t* L, int M) { return gptq_workspace_bytes_ex(L, M, nullptr); }

size_t gptq_workspace_bytes_max(const gptq_layer_t* L, int max_M) {
    size_t best = 0;
    for (int M = 1; M <= max_M; ++M) best = std::max(best, gptq_workspace_bytes_ex(L, M, nullptr));   // host arithmetic only
    return best;
}

int gptq_init(void) {
    hipError_t e = init_gemm_device();
    if (e == hipSuccess) e = init_gemv_device();
    if (e == hipSuccess) e = init_mlp_device();
    if (e == hipSuccess) e = init_gemm_mid_device();
    if (e == hipSuccess) e = init_gemv_tiled_device();
    if (e == hipSuccess) e = init_gemm_wide_sk_device();
    if (e == hipSuccess) e = init_gemm_rows_device();
    if (e == hipSuccess) e = init_gemm_panel_device();
    if (e != hipSuccess) return hip_fail(e, "gptq_init (hipFuncSetAttribute)");
    return GPTQ_OK;
}

int gptq_validate_g_idx(const int32_t* g_idx, int K, int G) {
    if (!g_idx) return fail(GPTQ_ERR_NULL, "g_idx is NULL");
    if (K <= 0 || G <= 0) return fail(GPTQ_ERR_SHAPE, "K (%d) and G (%d) must be > 0", K, G);
    for (int k = 0; k < K; ++k)
        if (g_idx[k] < 0 || g_idx[k] >= G)
            return fail(GPTQ_ERR_SHAPE, "g_idx[%d] = %d is outside [0, %d) (rows of scales / qzeros)", k, g_idx[k], G);
    return GPTQ_OK;
}

// The kernels see the workspace as (ticket header, body): the header is the caller's zeroed first 64 KiB and is NEVER relocated -- a call
// that nests another one (the unfused SILU_MUL epilogue) hands the inner call the same header and the rest of its body.
static int gemv_core(const gptq_layer_t* L, const void* x, void* out, int M, const WsView& wv, void* stream, const gptq_tuning_t* tune) {
    int rc = check_layer(L);
    if (rc) return rc;
    rc = check_io(x, out, M);
    if (rc) return rc;
    if (tune && tune->lanes_n && tune->lanes_n != 4 && tune->lanes_n != 8 && tune->lanes_n != 16 && tune->lanes_n != 64)   // = the header's list
        return fail(GPTQ_ERR_UNSUPPORTED, "tuning.lanes_n must be 4, 8, 16 or 64");
    if (tune && (tune->waves < 0 || tune->waves > 16)) return fail(GPTQ_ERR_UNSUPPORTED, "tuning.waves must be 1..16");
    GemvPlan pl = plan_gemv(*L, M, tune);
    if (L->epilogue != GPTQ_EPI_NONE && !pl.pair)
        return fail(GPTQ_ERR_UNSUPPORTED, "this GEMV kernel has no fused epilogue; call gptq_forward[_ex], which stages y in the workspace");
    if (tune && tune->path == 5 && !pl.mfma && !pl.mfmag)
        return fail(GPTQ_ERR_UNSUPPORTED, "matrix-core GEMV needs fp16 / bf16, groups of whole packing units (4-bit: a power-of-two group_size >= 8), "
                                          "16-column strips for 2/3/8-bit and no raw act-order g_idx");
    if (pl.mfma && pl.ln != 4 && (L->dtype != GPTQ_F16 || pl.mt > 4 || pl.use_seq || pl.pair || (pl.ln != 8 && pl.ln != 16)))
        return fail(GPTQ_ERR_UNSUPPORTED, "matrix-core GEMV: 32- / 64-column strips (lanes_n = 8 / 16) exist for plain fp16 layers at up to 4 rows only");
    if (!pl.mfma && !pl.mfmag && pl.ln != 4 && pl.ln != 16 && !(pl.ln == 8 && L->dtype == GPTQ_F32))
        return fail(GPTQ_ERR_UNSUPPORTED, "fp32-math GEMV: strips of 16 or 64 columns only (lanes_n = 4 / 16; fp32 layers also 8)");
    if (tune && (tune->path == 2 || tune->path == 4))
        return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = %d (round 1's LDS-staged / v_dot2 comparison GEMVs) was retired in round 6", tune->path);
    if (pl.workspace_bytes > 0 && wv.body_bytes < pl.workspace_bytes)
        return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + pl.workspace_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
    hipError_t e = launch_gemv(*L, pl, x, out, M, wv.body, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_gemv launch");
    return GPTQ_OK;
}
int gptq_gemv(const gptq_layer_t* L, const void* x, void* out, int M, void* ws, size_t ws_bytes, void* stream,
              const gptq_tuning_t* tune) {
    return gemv_core(L, x, out, M, split_ws(ws, ws_bytes), stream, tune);
}

static int gemm_core(const gptq_layer_t* L, const void* x, void* out, int M, const WsView& wv, void* stream, const gptq_tuning_t* tune) {
    int rc = check_layer(L);
    if (rc) return rc;
    rc = check_io(x, out, M);
    if (rc) return rc;
    if (L->epilogue != GPTQ_EPI_NONE)
        return fail(GPTQ_ERR_UNSUPPORTED, "the MFMA GEMM has no fused epilogue; call gptq_forward[_ex], which stages y in the workspace");
    GemmPlan pl = plan_gemm(*L, M, tune);
    if (!pl.supported)
        return fail(GPTQ_ERR_UNSUPPORTED,
                    "MFMA GEMM needs sequential or re-sequenced groups made of whole packing units (fp16/bf16: group_size %% 32 == 0) (bits=%d dtype=%d group_size=%d)",
                    L->bits, L->dtype, L->group_size);
    if (pl.workspace_bytes > 0 && wv.body_bytes < pl.workspace_bytes)
        return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + pl.workspace_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
    hipError_t e = launch_gemm(*L, pl, x, out, M, wv.header, wv.body, (hipStream_t)stream);
    if (e != hipSuccess)
        return hip_fail(e, pl.kg == 2 ? "gptq_gemm launch (this kernel needs > 64 KiB of LDS: was gptq_init() called on this device?)"
                                      : "gptq_gemm launch");
    return GPTQ_OK;
}
int gptq_gemm(const gptq_layer_t* L, const void* x, void* out, int M, void* ws, size_t ws_bytes, void* stream,
              const gptq_tuning_t* tune) {
    return gemm_core(L, x, out, M, split_ws(ws, ws_bytes), stream, tune);
}

static int stream_call(const gptq_layer_t* const* Ls, int n, const StreamPlan& sp, const void* x, void* const* outs, int M, const WsView& wv,
                       void* stream) {
    if (sp.partial_bytes > 0 && wv.body_bytes < sp.partial_bytes)
        return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + sp.partial_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
    hipError_t e = launch_stream(Ls, sp, x, outs, M, wv.header, wv.body, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq streamed GEMV launch (16 waves x U = 8 needs > 64 KiB of LDS: was gptq_init() called on this device?)");
    return GPTQ_OK;
}

static int tiled_call(const gptq_layer_t* const* Ls, int n, const void* x, void* const* outs, int M, const WsView& wv, void* stream, const gptq_tuning_t* tune,
                      const TiledPlan* planned = nullptr) {
    const TiledPlan tp = planned ? *planned : plan_tiled(Ls, n, M, tune);
    if (tp.partial_bytes > 0 && wv.body_bytes < tp.partial_bytes)
        return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + tp.partial_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
    hipError_t e = launch_tiled(Ls, tp, x, outs, M, wv.header, wv.body, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq strip-major decode launch (was gptq_init() called on this device?)");
    return GPTQ_OK;
}

static int forward_impl(const gptq_layer_t* L, const void* x, void* out, int M, const WsView& wv, void* stream,
                        const gptq_tuning_t* tune) {
    int rc = check_layer(L);
    if (rc) return rc;
    const size_t have = wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0;
    if (L->epilogue == GPTQ_EPI_NONE) {
        const gptq_layer_t* one[1] = {L};
        TiledPlan tp;
        if (want_tiled(one, 1, M, tune, &tp)) {
            rc = check_io(x, out, M);
            if (rc) return rc;
            void* outs[1] = {out};
            return tiled_call(one, 1, x, outs, M, wv, stream, tune, &tp);
        }
        if (tune && tune->path == 8)
            return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = 8: the decode-copy kernel needs M <= 8 and a plain or re-sequenced act-order 3/4/8-bit fp16/bf16 layer that carries qweight_tiled / qconst_tiled "
                                              "(gptq_prepack_decode; tiled_cols = %d)", GPTQ_STRIP_COLS);
    }
    if (tiled_pair_call(L, M, tune)) {                                            // SiLU * mul as the epilogue of the decode-copy kernel (round 5)
        rc = check_io(x, out, M);
        if (rc) return rc;
        const gptq_layer_t* one[1] = {L};
        void* outs[1] = {out};
        return tiled_call(one, 1, x, outs, M, wv, stream, tune);
    }
    if (L->epilogue != GPTQ_EPI_NONE && !fused_epilogue_ok(L, M, tune)) {
        rc = check_io(x, out, M);
        if (rc) return rc;
        const size_t yb = epi_scratch_bytes(L, M);
        if (wv.body_bytes < yb)
            return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", gptq_workspace_bytes_ex(L, M, tune), have);
        // y = [gate | up] at the front of the body; the inner call gets the SAME zeroed ticket header and the body behind y (a header carved
        // out of the body would sit on whatever an earlier, larger call left there: its tickets would never reach ksplit - 1)
        gptq_layer_t Lc = *L;
        Lc.epilogue = GPTQ_EPI_NONE;
        gptq_tuning_t local;
        rc = forward_impl(&Lc, x, wv.body, M, WsView{wv.header, (char*)wv.body + yb, wv.body_bytes - yb}, stream, inner_tuning(L, M, tune, &local));
        if (rc) return rc;
        hipError_t e = launch_silu_mul(wv.body, out, M, L->N, L->dtype, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "silu_mul launch");
        return GPTQ_OK;
    }
    if (want_stream(L, M, tune)) {
        rc = check_io(x, out, M);
        if (rc) return rc;
        const gptq_layer_t* one[1] = {L};
        void* outs[1] = {out};
        return stream_call(one, 1, plan_stream(one, 1, M, tune), x, outs, M, wv, stream);
    }
    {
        gptq_layer_t P;
        if (want_stream_seq(L, M, tune, &P)) {
            rc = check_io(x, out, M);
            if (rc) return rc;
            const gptq_layer_t* one[1] = {&P};
            void* outs[1] = {out};
            const StreamPlan sp = plan_stream(one, 1, M, nullptr);
            const size_t xb = xperm16_bytes(L, M);
            if (wv.body_bytes < xb + sp.partial_bytes)
                return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + xb + sp.partial_bytes, have);
            hipError_t e = launch_permute_columns(x, L->perm, M, L->K, L->dtype, wv.body, (hipStream_t)stream);
            if (e != hipSuccess) return hip_fail(e, "gptq_permute_columns launch");
            e = launch_stream(one, sp, wv.body, outs, M, wv.header, (char*)wv.body + xb, (hipStream_t)stream);
            if (e != hipSuccess) return hip_fail(e, "gptq streamed GEMV launch (was gptq_init() called on this device?)");
            return GPTQ_OK;
        }
    }
    if (tune && tune->path == 6)
        return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = 6: the streamed GEMV needs M <= 4, a plain 4-bit fp16/bf16 layer (no act-order, no epilogue) "
                                          "and a launch shape with rows-per-lane in {2, 4, 8} dividing the rows of a group");
    if (want_gemm(L, M, tune)) return gemm_core(L, x, out, M, wv, stream, tune);
    return gemv_core(L, x, out, M, wv, stream, tune);
}

int gptq_forward_ex(const gptq_layer_t* L, const void* x, void* out, int M, void* ws, size_t ws_bytes, void* stream,
                    const gptq_tuning_t* tune) {
    return forward_impl(L, x, out, M, split_ws(ws, ws_bytes), stream, tune);
}

size_t gptq_workspace_bytes_multi(const gptq_layer_t* const* layers, int n_layers, int M) {
    return gptq_workspace_bytes_multi_ex(layers, n_layers, M, nullptr);
}

// Batched decode of 2..4 layers sharing x in ONE gemm_stream64_kernel launch: by the planner's measured preference, or forced
// with tuning.path = 3 and tuning.reserved[2] = 4.
static bool want_stream64_multi(const gptq_layer_t* const* layers, int n, int M, const gptq_tuning_t* t) {
    if (n < 2 || n > 4 || M > 64) return false;
    const bool forced = t && t->path == 3 && t->reserved[2] == 4;
    if (t && t->path != 0 && !forced) return false;
    if (M <= 2 && !forced) return false;                         // up to 2 rows: the streamed GEMV (or layer by layer)
    const Stream64Plan sp = plan_stream64(layers, n, M, t);
    // 3..4 rows: only one unsplit round of 16-wave workgroups beats the streamed GEMV (tools/misc_bench.py, us, GEMV at M = 4 against this kernel
    // at M = 8: q|k|v, 192 strips: 12.0 against 11.2; gate|up, 344 strips in 8-wave workgroups: 17.5 against 18.8)
    if (M <= 4 && !forced && !(sp.ok && sp.ksplit == 1 && sp.strips_total <= 256)) return false;
    return sp.ok && (sp.pays || forced);
}

// 17 .. 128 rows, 2..4 layers sharing x in ONE gemm_mid_kernel launch: by the planner's measured preference, or forced with tuning.path = 3 and
// tuning.reserved[2] = 5.
static bool want_mid_multi(const gptq_layer_t* const* layers, int n, int M, const gptq_tuning_t* t) {
    if (n < 2 || n > 4 || M > 256) return false;
    const bool forced = t && t->path == 3 && t->reserved[2] == 5;
    if (t && t->path != 0 && !forced) return false;
    const MidPlan mp = plan_mid(layers, n, M, t);
    return mp.ok && (mp.pays || forced);
}

// 33+ rows of a group whose weights are mostly (>= 3/4) in layers the panel kernel takes with well-filled rounds: the layers run ONE BY ONE -- the multi-layer
// forms of the rows / mid / stream64 kernels were fitted before that kernel existed and lose to it (tools/multi_rows_sweep.py, profiles/r06_multi_rows_sweep.log, group
// call -> one by one: 8B gate|up at 64 rows 45.7 -> 32.8 us, 70B q|k|v at 128 rows 59.5 -> 46.9, 70B TP8 gate|up at 128 rows 42.0 -> 37.9; 13B q|k|v at 128 rows --
// 160 tiles of 256 per layer -- keeps its group launch: 45.0 against 52.8)
static bool separate_panels_pay(const gptq_layer_t* const* layers, int n, int M) {
    if (n < 2 || M < 33) return false;
    size_t all = 0, paneled = 0;
    for (int i = 0; i < n; ++i) {
        const size_t kn = (size_t)layers[i]->K * layers[i]->N;
        all += kn;
        if (layers[i]->epilogue == GPTQ_EPI_NONE && want_gemm(layers[i], M, nullptr) && panel_pays_filled(*layers[i], M)) paneled += kn;
    }
    return paneled * 4 >= all * 3;
}

size_t gptq_workspace_bytes_multi_ex(const gptq_layer_t* const* layers, int n_layers, int M, const gptq_tuning_t* tune) {
    if (!layers || n_layers <= 0 || M <= 0) return 0;
    for (int i = 0; i < n_layers; ++i)
        if (check_layer(layers[i]) != GPTQ_OK) return 0;
    size_t need = 0;
    if (want_tiled(layers, n_layers, M, tune) && (n_layers >= 2 || (tune && tune->path == 8))) {
        const TiledPlan tp = plan_tiled(layers, n_layers, M, tune);
        return tp.partial_bytes ? WS_HEADER_BYTES + tp.partial_bytes : 0;
    }
    const bool separate = !tune && separate_panels_pay(layers, n_layers, M);
    if (!separate && n_layers >= 2 && ((tune && tune->path == 3 && tune->reserved[3] == GPTQ_LAB_VARIANT_ROWS_ON && rows_multi_ok(layers, n_layers, M)) || (!tune && rows_multi_pays(layers, n_layers, M))) &&
        plan_rows_multi(layers, n_layers, M, nullptr).ok)
        return layers[0]->g_idx ? WS_HEADER_BYTES + xperm16_bytes(layers[0], M) : 0;      // gemm_rows.hip over all layers: nothing but the permuted x of act-order layers
    if (separate) {
        for (int i = 0; i < n_layers; ++i) need = std::max(need, gptq_workspace_bytes_ex(layers[i], M, nullptr));
        return need;
    }
    if (want_mid_multi(layers, n_layers, M, tune)) {
        const MidPlan mp = plan_mid(layers, n_layers, M, tune);
        return mp.partial_bytes ? WS_HEADER_BYTES + mp.partial_bytes : 0;
    }
    if (want_stream64_multi(layers, n_layers, M, tune)) {
        const Stream64Plan sp = plan_stream64(layers, n_layers, M, tune);
        return sp.partial_bytes ? WS_HEADER_BYTES + sp.partial_bytes : 0;
    }
    if (n_layers <= 4) {
        const StreamPlan sp = plan_stream(layers, n_layers, M, tune);
        if (sp.ok && multi_preferred(layers, n_layers, M)) return sp.partial_bytes ? WS_HEADER_BYTES + sp.partial_bytes : 0;
    }
    for (int i = 0; i < n_layers; ++i) need = std::max(need, gptq_workspace_bytes_ex(layers[i], M, nullptr));
    return need;
}

int gptq_forward_multi(const gptq_layer_t* const* layers, int n_layers, const void* x, void* const* outs, int M, void* ws,
                       size_t ws_bytes, void* stream) {
    return gptq_forward_multi_ex(layers, n_layers, x, outs, M, ws, ws_bytes, stream, nullptr);
}

static int forward_multi_core(const gptq_layer_t* const* layers, int n_layers, const void* x, void* const* outs, int M, const WsView& wv,
                              void* stream, const gptq_tuning_t* tune) {
    if (!layers || !outs) return fail(GPTQ_ERR_NULL, "layers/outs must be non-NULL");
    if (n_layers <= 0) return fail(GPTQ_ERR_SHAPE, "n_layers must be > 0, got %d", n_layers);
    for (int i = 0; i < n_layers; ++i) {
        int rc = check_layer(layers[i]);
        if (rc) return rc;
        rc = check_io(x, outs[i], M);
        if (rc) return rc;
        if (layers[i]->K != layers[0]->K) return fail(GPTQ_ERR_SHAPE, "layers of one gptq_forward_multi call read the same x: in_features %d != %d", layers[i]->K, layers[0]->K);
        if (layers[i]->dtype != layers[0]->dtype) return fail(GPTQ_ERR_UNSUPPORTED, "layers of one gptq_forward_multi call share the dtype of x");
    }
    // decode rows on layers that carry the decode copy: one launch over the strips of all layers (tools/tiled_sweep.py, M = 4, us: q|k|v 8.6 against
    // 11.1 for the batched-decode kernel, gate|up 12.9 against 18.9)
    {
        TiledPlan tp;
        // (no byte cap on the group for this kernel -- multi_preferred's 128 MB was measured on the round-2 kernels: 70B gate|up, 235 MB, M = 1 / 4 one by one
        // 48.5 / 57.3 us, ONE decode-copy launch 42.2 / 50.9: profiles/r06_multi_geom_sweep.log)
        if (want_tiled(layers, n_layers, M, tune, &tp) && (n_layers >= 2 || (tune && tune->path == 8)))
            return tiled_call(layers, n_layers, x, outs, M, wv, stream, tune, &tp);
    }
    if (tune && tune->path == 8) return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = 8: these layers do not fit one decode-copy launch (1..4 layers of one packing with qweight_tiled, all plain or all act-order, M <= 8)");
    // 5 .. 128 rows on layers that carry the decode copy: the exchange-free kernel over the strip groups of all layers (gemm_rows.hip); act-order layers of
    // ONE order (the same perm pointer: QuantLinear.share_act_order) read one permuted x
    const bool rows_forced = tune && tune->path == 3 && tune->reserved[3] == GPTQ_LAB_VARIANT_ROWS_ON && rows_multi_ok(layers, n_layers, M);      // lab knob 50
    const bool separate = !tune && separate_panels_pay(layers, n_layers, M);
    if (n_layers >= 2 && (rows_forced || (!tune && !separate && rows_multi_pays(layers, n_layers, M)))) {
        const RowsPlan rp = plan_rows_multi(layers, n_layers, M, nullptr);
        if (rp.ok) {
            const void* xin = x;
            if (layers[0]->g_idx) {
                const size_t xb = xperm16_bytes(layers[0], M);
                if (wv.body_bytes < xb) return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + xb, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
                hipError_t e = launch_permute_rows16(x, layers[0]->perm, M, layers[0]->K, wv.body, (hipStream_t)stream, false);
                if (e != hipSuccess) return hip_fail(e, "gptq x permute launch");
                xin = wv.body;
            }
            hipError_t e = launch_gemm_rows_multi(layers, n_layers, rp, xin, outs, M, (hipStream_t)stream);
            if (e != hipSuccess) return hip_fail(e, "gptq batched-decode launch (gemm_rows; was gptq_init() called on this device?)");
            return GPTQ_OK;
        }
    }
    if (!separate && want_mid_multi(layers, n_layers, M, tune)) {
        const MidPlan mp = plan_mid(layers, n_layers, M, tune);
        if (mp.partial_bytes > 0 && wv.body_bytes < mp.partial_bytes)
            return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + mp.partial_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
        hipError_t e = launch_mid(layers, mp, x, outs, M, wv.header, wv.body, nullptr, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "gptq 17..128-row launch (needs > 64 KiB of LDS: was gptq_init() called on this device?)");
        return GPTQ_OK;
    }
    if (!separate && want_stream64_multi(layers, n_layers, M, tune)) {
        const Stream64Plan sp = plan_stream64(layers, n_layers, M, tune);
        if (sp.partial_bytes > 0 && wv.body_bytes < sp.partial_bytes)
            return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + sp.partial_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
        hipError_t e = launch_stream64(layers, sp, x, outs, M, wv.header, wv.body, nullptr, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "gptq batched-decode launch (needs > 64 KiB of LDS: was gptq_init() called on this device?)");
        return GPTQ_OK;
    }
    if (n_layers <= 4 && M <= 4) {
        const StreamPlan sp = plan_stream(layers, n_layers, M, tune);
        if (sp.ok && multi_preferred(layers, n_layers, M)) return stream_call(layers, n_layers, sp, x, outs, M, wv, stream);
        if (tune && tune->path == 6) return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = 6: these layers / this launch shape do not fit the streamed GEMV");
    }
    if (tune && tune->path == 3 && tune->reserved[2] == 4)
        return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = 3 / reserved[2] = 4: these layers do not fit one batched-decode launch (2..4 plain 4-bit layers, M <= 64)");
    if (tune && tune->path == 3 && tune->reserved[2] == 5)
        return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = 3 / reserved[2] = 5: these layers do not fit one gemm_mid_kernel launch (2..4 plain 4-bit layers, N %% 64 == 0, M <= 256)");
    // Act-order layers that share ONE perm -- q / k / v and gate / up of a GPTQ checkpoint do: the activation order comes from the Hessian of the
    // layers' common input (the reference's fused q/k/v caller relies on it: fused_llama_attn.py:188 hands the kernels q_proj's order for all three) --
    // read ONE permuted x: the first layer's GEMM call permutes into the head of the workspace, the others find it there.  The caller says so by
    // passing the same `perm` pointer (QuantLinear.share_act_order / forward_multi do that once the g_idx tensors compared equal).
    if (n_layers >= 2 && !tune && layers[0]->perm && layers[0]->qweight_seq && layers[0]->g_idx) {
        GemmPlan pls[4];
        bool shared = n_layers <= 4;
        for (int i = 0; shared && i < n_layers; ++i) {
            const gptq_layer_t* L = layers[i];
            shared = L->perm == layers[0]->perm && L->qweight_seq && L->g_idx && L->epilogue == GPTQ_EPI_NONE && L->K == layers[0]->K &&
                     L->dtype == layers[0]->dtype && L->dtype != GPTQ_F32 && want_gemm(L, M, nullptr);
            if (!shared) break;
            pls[i] = plan_gemm(*L, M, nullptr);
            shared = pls[i].supported && pls[i].use_seq && !pls[i].f32 && pls[i].xslot == pls[0].xslot && pls[i].xnat == pls[0].xnat && pls[i].xperm_bytes == pls[0].xperm_bytes &&
                     pls[i].workspace_bytes <= wv.body_bytes;
        }
        if (shared) {
            for (int i = 0; i < n_layers; ++i) {
                hipError_t e = launch_gemm(*layers[i], pls[i], x, outs[i], M, wv.header, wv.body, (hipStream_t)stream, i > 0);
                if (e != hipSuccess) return hip_fail(e, "gptq_gemm launch (shared permuted x)");
            }
            return GPTQ_OK;
        }
    }
    for (int i = 0; i < n_layers; ++i) {           // anything the one-launch kernel does not cover: the same result, layer by layer
        int rc = forward_impl(layers[i], x, outs[i], M, wv, stream, nullptr);
        if (rc) return rc;
    }
    return GPTQ_OK;
}

int gptq_forward_multi_ex(const gptq_layer_t* const* layers, int n_layers, const void* x, void* const* outs, int M, void* ws,
                          size_t ws_bytes, void* stream, const gptq_tuning_t* tune) {
    return forward_multi_core(layers, n_layers, x, outs, M, split_ws(ws, ws_bytes), stream, tune);
}

// ---- fused gated MLP (mlp.hip) -----------------------------------------------------------------------------------------
static int check_mlp(const gptq_layer_t* gate, const gptq_layer_t* up, const gptq_layer_t* down) {
    int rc = check_layer(gate);
    if (rc) return rc;
    if ((rc = check_layer(up))) return rc;
    if ((rc = check_layer(down))) return rc;
    if (gate->K != up->K || gate->N != up->N)
        return fail(GPTQ_ERR_SHAPE, "gate (%d -> %d) and up (%d -> %d) must have the same shape", gate->K, gate->N, up->K, up->N);
    if (down->K != gate->N)
        return fail(GPTQ_ERR_SHAPE, "down reads the intermediate activation: its in_features (%d) must be gate's out_features (%d)", down->K, gate->N);
    if (gate->dtype != up->dtype || gate->dtype != down->dtype) return fail(GPTQ_ERR_UNSUPPORTED, "the three layers of an MLP share the dtype of x");
    if (gate->epilogue != GPTQ_EPI_NONE || up->epilogue != GPTQ_EPI_NONE || down->epilogue != GPTQ_EPI_NONE)
        return fail(GPTQ_ERR_UNSUPPORTED, "gptq_mlp_forward takes plain layers (the SiLU * mul is its own)");
    return GPTQ_OK;
}
static size_t mlp_stage_bytes(const gptq_layer_t* gate, int M) { return ((size_t)M * gate->N * dtype_size(gate->dtype) + 255) / 256 * 256; }

size_t gptq_workspace_bytes_mlp(const gptq_layer_t* gate, const gptq_layer_t* up, const gptq_layer_t* down, int M) {
    return gptq_workspace_bytes_mlp_ex(gate, up, down, M, nullptr);
}

size_t gptq_workspace_bytes_mlp_ex(const gptq_layer_t* gate, const gptq_layer_t* up, const gptq_layer_t* down, int M, const gptq_tuning_t* tune) {
    (void)tune;
    if (check_mlp(gate, up, down) != GPTQ_OK || M <= 0) return 0;
    const gptq_layer_t* gu[2] = {gate, up};
    size_t inner = gptq_workspace_bytes_multi_ex(gu, 2, M, nullptr);
    inner = std::max(inner, gptq_workspace_bytes_ex(down, M, nullptr));
    inner = inner > WS_HEADER_BYTES ? inner - WS_HEADER_BYTES : 0;
    return WS_HEADER_BYTES + 2 * mlp_stage_bytes(gate, M) + inner;
}

// down of gptq_mlp_forward: does its call at M rows run an MFMA GEMM that reads x permuted in NATURAL order of its re-sequenced rows (GemmPlan.xnat)?  Then the
// SiLU * mul pass writes that permuted x itself.  (Decode rows gather inside the kernel, fp32 and the checkpoint-row kernels have their own orders: unchanged.)
static bool mlp_fused_permute(const gptq_layer_t* down, int M, const gptq_tuning_t* tune) {
    if (tune && tune->reserved[3] == GPTQ_LAB_VARIANT_MLP_TWO_PASSES) return false;      // lab (bench.py: the two passes of round 5, interleaved with the default)
    if (!down->g_idx || !down->perm || !down->qweight_seq || down->epilogue != GPTQ_EPI_NONE) return false;
    if (!silu_mul2_permute_ok(down->K, down->dtype)) return false;
    const gptq_layer_t* one[1] = {down};
    gptq_layer_t P;
    if (want_tiled(one, 1, M, nullptr) || want_stream(down, M, nullptr) || want_stream_seq(down, M, nullptr, &P) || !want_gemm(down, M, nullptr)) return false;      // (forward_impl's order)
    const GemmPlan pl = plan_gemm(*down, M, nullptr);
    return pl.supported && pl.use_seq && pl.xnat && !pl.f32 && pl.xperm_bytes > 0;
}

int gptq_mlp_forward(const gptq_layer_t* gate, const gptq_layer_t* up, const gptq_layer_t* down, const void* x, void* out, int M,
                     void* ws, size_t ws_bytes, void* stream) {
    return gptq_mlp_forward_ex(gate, up, down, x, out, M, ws, ws_bytes, stream, nullptr);
}

int gptq_mlp_forward_ex(const gptq_layer_t* gate, const gptq_layer_t* up, const gptq_layer_t* down, const void* x, void* out, int M,
                        void* ws, size_t ws_bytes, void* stream, const gptq_tuning_t* tune) {
    int rc = check_mlp(gate, up, down);
    if (rc) return rc;
    if ((rc = check_io(x, out, M))) return rc;
    const WsView wv = split_ws(ws, ws_bytes);
    const size_t have = wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0;
    if (tune && tune->path != 0)
        return fail(GPTQ_ERR_UNSUPPORTED, "gptq_mlp_forward_ex takes no path override (tuning.path = %d): the one-launch persistent kernel of round 3 is a lab now "
                                          "(tools/lab/mlp_ring.hip), not part of this library", tune->path);
    // Three steps -- gate and up through the multi-layer entry point (ONE launch for decode rows) into two staging buffers at the front of the body,
    // SiLU * mul in place, down -- each inner call on the ONE ticket header and the body behind the staging buffers.
    const size_t sb = mlp_stage_bytes(gate, M);
    if (wv.body_bytes < 2 * sb)
        return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", gptq_workspace_bytes_mlp_ex(gate, up, down, M, tune), have);
    void* hg = wv.body;
    void* hu = (char*)wv.body + sb;
    const WsView inner{wv.header, (char*)wv.body + 2 * sb, wv.body_bytes - 2 * sb};
    const gptq_layer_t* gu[2] = {gate, up};
    void* outs[2] = {hg, hu};
    if ((rc = forward_multi_core(gu, 2, x, outs, M, inner, stream, nullptr))) return rc;
    // An act-order `down` whose kernel reads x permuted in natural order (the decode-copy GEMMs: panel / rows / stream-K / wide tiles): SiLU * mul and that
    // permute are ONE pass into the spot down's own pre-pass would have filled (round 6; the reference's fused MLP, fused_llama_mlp.py:131-306)
    if (mlp_fused_permute(down, M, tune)) {
        const GemmPlan pl = plan_gemm(*down, M, nullptr);
        if (pl.workspace_bytes > inner.body_bytes)
            return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", gptq_workspace_bytes_mlp_ex(gate, up, down, M, tune), have);
        hipError_t e = launch_silu_mul2_permute(hg, hu, down->perm, M, down->K, down->dtype, inner.body, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "silu_mul + permute launch (was gptq_init() called on this device?)");
        e = launch_gemm(*down, pl, inner.body, out, M, inner.header, inner.body, (hipStream_t)stream, /*x_permuted=*/true);
        if (e != hipSuccess) return hip_fail(e, "gptq_gemm launch (down projection of gptq_mlp_forward)");
        return GPTQ_OK;
    }
    hipError_t e = launch_silu_mul2(hg, hu, hg, (size_t)M * gate->N, gate->dtype, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "silu_mul launch");
    return forward_impl(down, hg, out, M, inner, stream, nullptr);
}

int gptq_describe_mlp_plan(const gptq_layer_t* gate, const gptq_layer_t* up, const gptq_layer_t* down, int M, const gptq_tuning_t* tune, char* out,
                           size_t out_bytes) {
    if (!out || out_bytes == 0) return fail(GPTQ_ERR_NULL, "out must be non-NULL");
    out[0] = 0;
    int rc = check_mlp(gate, up, down);
    if (rc) return rc;
    if (M <= 0) return fail(GPTQ_ERR_SHAPE, "M must be > 0, got %d", M);
    if (mlp_fused_permute(down, M, tune)) snprintf(out, out_bytes, "kernel=unfused launches=3 steps=forward_multi(gate,up)|silu_mul+permute|gemm(down) down_permute=fused");
    else {
        const gptq_layer_t* one[1] = {down};
        const char* dp = !down->g_idx ? "none" : (want_tiled(one, 1, M, nullptr) ? "in_kernel" : "own_pass");      // (decode rows: the decode kernel gathers x through perm itself)
        snprintf(out, out_bytes, "kernel=unfused launches=3+ steps=forward_multi(gate,up)|silu_mul|forward(down) down_permute=%s", dp);
    }
    return GPTQ_OK;
}

int gptq_forward(const gptq_layer_t* L, const void* x, void* out, int M, void* ws, size_t ws_bytes, void* stream) {
    return gptq_forward_ex(L, x, out, M, ws, ws_bytes, stream, nullptr);
}

int gptq_dequant(const gptq_layer_t* L, void* W_out, void* stream) {
    int rc = check_layer(L);
    if (rc) return rc;
    if (!W_out) return fail(GPTQ_ERR_NULL, "W_out is NULL");
    hipError_t e = launch_dequant(*L, W_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_dequant launch");
    return GPTQ_OK;
}

int gptq_unpack_weights(const uint32_t* qweight, int K, int N, int bits, uint8_t* w_out, void* stream) {
    if (!qweight || !w_out) return fail(GPTQ_ERR_NULL, "qweight/w_out must be non-NULL");
    if (!bits_ok(bits)) return fail(GPTQ_ERR_UNSUPPORTED, "Only 2,3,4,8 bits are supported. (got %d)", bits);
    if (K <= 0 || N <= 0 || K % 32 || N % 32) return fail(GPTQ_ERR_SHAPE, "K (%d), N (%d) must be positive multiples of 32", K, N);
    hipError_t e = launch_unpack_weights(qweight, K, N, bits, w_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_unpack_weights launch");
    return GPTQ_OK;
}

int gptq_unpack_zeros(const uint32_t* qzeros, int G, int N, int bits, int zero_mode, int32_t* z_out, void* stream) {
    if (!qzeros || !z_out) return fail(GPTQ_ERR_NULL, "qzeros/z_out must be non-NULL");
    if (!bits_ok(bits)) return fail(GPTQ_ERR_UNSUPPORTED, "Only 2,3,4,8 bits are supported. (got %d)", bits);
    if (G <= 0 || N <= 0 || N % 32) return fail(GPTQ_ERR_SHAPE, "G (%d) must be > 0 and N (%d) a positive multiple of 32", G, N);
    hipError_t e = launch_unpack_zeros(qzeros, G, N, bits, zero_mode, z_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_unpack_zeros launch");
    return GPTQ_OK;
}

int gptq_pack_weights(const void* W, const void* scale_in, const void* zero_in, const int32_t* g_idx, int K, int N,
                      int bits, int group_size, int w_dtype, int qparam_dtype, uint32_t* qweight_out, void* scales_out,
                      void* stream) {
    if (!W || !scale_in || !zero_in || !qweight_out) return fail(GPTQ_ERR_NULL, "W/scale_in/zero_in/qweight_out must be non-NULL");
    if (!bits_ok(bits)) return fail(GPTQ_ERR_UNSUPPORTED, "Only 2,3,4,8 bits are supported. (got %d)", bits);
    if (!dtype_ok(w_dtype) || !dtype_ok(qparam_dtype)) return fail(GPTQ_ERR_UNSUPPORTED, "unsupported dtype enum");
    if (K <= 0 || N <= 0 || K % 32 || N % 32 || group_size <= 0)
        return fail(GPTQ_ERR_SHAPE, "K (%d), N (%d) must be positive multiples of 32 and group_size (%d) > 0", K, N, group_size);
    hipError_t e = launch_pack_weights(W, scale_in, zero_in, g_idx, K, N, bits, group_size, w_dtype, qparam_dtype,
                                       qweight_out, scales_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_pack_weights launch");
    return GPTQ_OK;
}

int gptq_pack_zeros(const void* zero_in, int G, int N, int bits, int qparam_dtype, uint32_t* qzeros_out, void* stream) {
    if (!zero_in || !qzeros_out) return fail(GPTQ_ERR_NULL, "zero_in/qzeros_out must be non-NULL");
    if (!bits_ok(bits)) return fail(GPTQ_ERR_UNSUPPORTED, "Only 2,3,4,8 bits are supported. (got %d)", bits);
    if (!dtype_ok(qparam_dtype)) return fail(GPTQ_ERR_UNSUPPORTED, "unsupported dtype enum");
    if (G <= 0 || N <= 0 || N % 32) return fail(GPTQ_ERR_SHAPE, "G (%d) must be > 0 and N (%d) a positive multiple of 32", G, N);
    hipError_t e = launch_pack_zeros(zero_in, G, N, bits, qparam_dtype, qzeros_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_pack_zeros launch");
    return GPTQ_OK;
}

int gptq_make_sequential(const int32_t* g_idx, int K, int group_size, int32_t* perm_out, int* uniform_out) {
    if (!g_idx || !perm_out) return fail(GPTQ_ERR_NULL, "g_idx/perm_out must be non-NULL");
    if (K <= 0 || group_size <= 0) return fail(GPTQ_ERR_SHAPE, "K (%d) and group_size (%d) must be > 0", K, group_size);
    int gmax = 0;
    for (int k = 0; k < K; ++k) {
        if (g_idx[k] < 0) return fail(GPTQ_ERR_SHAPE, "g_idx[%d] = %d is negative", k, g_idx[k]);
        gmax = std::max(gmax, g_idx[k]);
    }
    // stable counting sort by group (same ordering rule as Q4Matrix::make_sequential)
    std::vector<int> start((size_t)gmax + 2, 0);
    for (int k = 0; k < K; ++k) start[(size_t)g_idx[k] + 1]++;
    for (int g = 0; g <= gmax; ++g) start[(size_t)g + 1] += start[g];
    std::vector<int> cursor(start.begin(), start.end() - 1);
    for (int k = 0; k < K; ++k) perm_out[cursor[g_idx[k]]++] = k;
    if (uniform_out) {
        int uni = 1;
        for (int i = 0; i < K && uni; ++i) uni = (g_idx[perm_out[i]] == i / group_size);
        *uniform_out = uni;
    }
    return GPTQ_OK;
}

int gptq_resequence_qweight(const uint32_t* qweight, const int32_t* perm, int K, int N, int bits, uint32_t* out, void* stream) {
    if (!qweight || !perm || !out) return fail(GPTQ_ERR_NULL, "qweight/perm/out must be non-NULL");
    if (!bits_ok(bits)) return fail(GPTQ_ERR_UNSUPPORTED, "Only 2,3,4,8 bits are supported. (got %d)", bits);
    if (K <= 0 || N <= 0 || K % 32 || N % 32) return fail(GPTQ_ERR_SHAPE, "K (%d), N (%d) must be positive multiples of 32", K, N);
    hipError_t e = launch_resequence(qweight, perm, K, N, bits, out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_resequence_qweight launch");
    return GPTQ_OK;
}

// the layer as the decode copy sees it: plain 3/4/8-bit, source rows = qweight_seq when the layer has one
static int decode_copy_source(const gptq_layer_t* L, gptq_layer_t* S) {
    int rc = check_layer(L);
    if (rc) return rc;
    *S = *L;
    S->qweight_tiled = (const uint32_t*)(uintptr_t)16;       // placeholders: tiled_layer_ok() only asks whether the copy COULD exist
    S->qconst_tiled = (const void*)(uintptr_t)16;
    S->tiled_cols = GPTQ_STRIP_COLS;
    // a plain [gate | up] layer with the fused epilogue gets a copy too (round 5: its decode kernel pairs strip s of the two halves); act-order ones do not
    if (L->epilogue != GPTQ_EPI_NONE && (L->g_idx != nullptr || L->N % (2 * GPTQ_STRIP_COLS) != 0)) S->epilogue = -1;
    if (S->epilogue == -1 || !tiled_layer_ok(*S))
        return fail(GPTQ_ERR_UNSUPPORTED, "the decode copy needs a 2-, 3-, 4- or 8-bit fp16/bf16 layer (2 bits: no fused epilogue), group_size a power-of-two multiple of 32 (16 at 8 bits) or >= K, "
                                          "and for act-order layers qweight_seq + perm (bits=%d dtype=%d group_size=%d)", L->bits, L->dtype, L->group_size);
    return GPTQ_OK;
}

int gptq_prepack_decode_bytes(const gptq_layer_t* L, size_t* tiled_bytes, size_t* const_bytes) {
    if (tiled_bytes) *tiled_bytes = 0;
    if (const_bytes) *const_bytes = 0;
    if (!tiled_bytes || !const_bytes) return fail(GPTQ_ERR_NULL, "tiled_bytes/const_bytes must be non-NULL");
    gptq_layer_t S;
    if (int rc = decode_copy_source(L, &S)) return rc;
    *tiled_bytes = tiled_weight_bytes(*L);
    *const_bytes = tiled_const_bytes(*L);
    return GPTQ_OK;
}

int gptq_prepack_decode(const gptq_layer_t* L, uint32_t* tiled_out, void* const_out, void* stream) {
    if (!tiled_out || !const_out) return fail(GPTQ_ERR_NULL, "qweight_tiled_out/qconst_tiled_out must be non-NULL");
    gptq_layer_t S;
    if (int rc = decode_copy_source(L, &S)) return rc;
    const uint32_t* src = L->qweight_seq ? L->qweight_seq : L->qweight;
    if (tiled_out == src || tiled_out == L->qweight) return fail(GPTQ_ERR_UNSUPPORTED, "gptq_prepack_decode does not work in place (the checkpoint tensors are never rewritten)");
    hipError_t e = launch_prepack_decode(src, L->qzeros, L->scales, L->K, L->N, L->bits, L->group_size, L->zero_mode, tiled_out, const_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_prepack_decode launch");
    return GPTQ_OK;
}

int gptq_unprepack_decode(const uint32_t* qweight_tiled, int K, int N, int bits, uint32_t* qweight_out, void* stream) {
    if (!qweight_tiled || !qweight_out) return fail(GPTQ_ERR_NULL, "qweight_tiled/qweight_out must be non-NULL");
    if (bits != 2 && bits != 3 && bits != 4 && bits != 8) return fail(GPTQ_ERR_UNSUPPORTED, "the decode copy exists for 2-, 3-, 4- and 8-bit layers (got %d)", bits);
    if (K <= 0 || N <= 0 || K % 32 || N % GPTQ_STRIP_COLS) return fail(GPTQ_ERR_SHAPE, "K (%d) must be a positive multiple of 32 and N (%d) of %d", K, N, GPTQ_STRIP_COLS);
    if ((const void*)qweight_tiled == (const void*)qweight_out) return fail(GPTQ_ERR_UNSUPPORTED, "gptq_unprepack_decode does not work in place");
    hipError_t e = launch_unprepack_decode(qweight_tiled, K, N, bits, qweight_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_unprepack_decode launch");
    return GPTQ_OK;
}

int gptq_permute_columns(const void* x, const int32_t* perm, int M, int K, int dtype, void* x_out, void* stream) {
    if (!x || !perm || !x_out) return fail(GPTQ_ERR_NULL, "x/perm/x_out must be non-NULL");
    if (!dtype_ok(dtype)) return fail(GPTQ_ERR_UNSUPPORTED, "unsupported dtype enum %d", dtype);
    if (M <= 0 || K <= 0) return fail(GPTQ_ERR_SHAPE, "M (%d) and K (%d) must be > 0", M, K);
    hipError_t e = launch_permute_columns(x, perm, M, K, dtype, x_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_permute_columns launch");
    return GPTQ_OK;
}

int gptq_describe_plan(const gptq_layer_t* L, int M, const gptq_tuning_t* tune, char* out, size_t out_bytes) {
    if (!out || out_bytes == 0) return fail(GPTQ_ERR_NULL, "out must be non-NULL");
    out[0] = 0;
    int rc = check_layer(L);
    if (rc) return rc;
    if (M <= 0) return fail(GPTQ_ERR_SHAPE, "M must be > 0, got %d", M);
    if (tune && (tune->path == 2 || tune->path == 4)) return fail(GPTQ_ERR_UNSUPPORTED, "tuning.path = %d was retired in round 6", tune->path);
    if (tiled_pair_call(L, M, tune)) {
        const gptq_layer_t* one[1] = {L};
        const TiledPlan tp = plan_tiled(one, 1, M, tune);
        snprintf(out, out_bytes, "path=gemv kernel=strips ln=4 waves=%d u=%d ksplit=%d mt=%d strips=%d pair=1 perm=0 epilogue=fused", tp.waves, tp.u, tp.ksplit, tp.mt,
                 tp.strips_total);
        return GPTQ_OK;
    }
    const bool unfused_epilogue = L->epilogue != GPTQ_EPI_NONE && !fused_epilogue_ok(L, M, tune);
    gptq_layer_t Lc = *L;
    gptq_tuning_t local;
    if (unfused_epilogue) {
        Lc.epilogue = GPTQ_EPI_NONE;
        tune = inner_tuning(L, M, tune, &local);
    }
    gptq_layer_t Pseq;
    const gptq_layer_t* one_t[1] = {&Lc};
    if (!unfused_epilogue && Lc.epilogue == GPTQ_EPI_NONE && want_tiled(one_t, 1, M, tune)) {
        const TiledPlan tp = plan_tiled(one_t, 1, M, tune);
        snprintf(out, out_bytes, "path=gemv kernel=strips ln=4 waves=%d u=%d ksplit=%d mt=%d strips=%d pair=0 perm=0 epilogue=none", tp.waves, tp.u, tp.ksplit, tp.mt,
                 tp.strips_total);
    } else if (!unfused_epilogue && want_stream_seq(L, M, tune, &Pseq)) {
        const gptq_layer_t* one[1] = {&Pseq};
        const StreamPlan sp = plan_stream(one, 1, M, nullptr);
        snprintf(out, out_bytes, "path=gemv kernel=stream ln=%d waves=%d u=%d ksplit=%d mt=%d strips=%d pair=0 perm=2 epilogue=none", sp.ln, sp.waves,
                 sp.u, sp.ksplit, sp.mt, sp.strips_total);
    } else if (want_stream(&Lc, M, tune)) {
        const gptq_layer_t* one[1] = {&Lc};
        const StreamPlan sp = plan_stream(one, 1, M, tune);
        snprintf(out, out_bytes, "path=gemv kernel=stream ln=%d waves=%d u=%d ksplit=%d mt=%d strips=%d pair=0 perm=0 epilogue=none", sp.ln, sp.waves,
                 sp.u, sp.ksplit, sp.mt, sp.strips_total);
    } else if (want_gemm(&Lc, M, tune)) {
        const GemmPlan g = plan_gemm(Lc, M, tune);
        const char* kern = g.f32 ? "f32_mfma" : g.rows ? "rows" : g.panel ? "panel" : g.wsk ? "wide_sk" : g.wide ? (g.wide_tiled ? "wide_copy" : "wide") : g.mid ? "mid" : (g.stream64 ? "stream64" : (g.strip16 ? "strip16" : (g.skinny ? "skinny64" : "tiled")));
        snprintf(out, out_bytes, "path=gemm kernel=%s mt=%d bk=%d kg=%d ksplit=%d tiles=%dx%d tail=%d tail_slices=%d perm=%d dma=%d waves=%d u=%d epilogue=%s", kern, g.mt, g.bk,
                 g.kg == 2 ? 2 : 1, g.ksplit, g.nbm, g.nbn, g.tail, g.tail ? (1 << g.tail_lg) : 1, g.use_seq ? 1 : 0, (g.glds || g.stream64 || g.mid) ? 1 : 0, g.waves, g.u,
                 unfused_epilogue ? "separate" : "none");
    } else {
        const GemvPlan v = plan_gemv(Lc, M, tune);
        const char* kern = v.mfma ? "mfma" : (v.mfmag ? "mfma_generic" : "generic");
        snprintf(out, out_bytes, "path=gemv kernel=%s ln=%d waves=%d u=%d ksplit=%d mt=%d strips=%d pair=%d perm=%d epilogue=%s%s", kern, v.ln,
                 v.waves, v.u, v.ksplit, v.mt, v.strips, v.pair ? 1 : 0, v.xperm ? 2 : (v.use_seq ? 1 : 0),
                 v.pair ? "fused" : (unfused_epilogue ? "separate" : "none"), v.magic ? " deq=magic" : "");
    }
    return GPTQ_OK;
}

static int awq_shape_check(int K, int N, int group_size) {
    if (K <= 0 || N <= 0 || group_size <= 0) return fail(GPTQ_ERR_SHAPE, "K (%d), N (%d), group_size (%d) must be > 0", K, N, group_size);
    if (K % 8 || N % 8) return fail(GPTQ_ERR_SHAPE, "K (%d) and N (%d) must be multiples of 8 (4-bit words)", K, N);
    if (K % group_size) return fail(GPTQ_ERR_SHAPE, "K (%d) must be a multiple of group_size (%d)", K, group_size);
    return GPTQ_OK;
}

int gptq_awq_unpack(const uint32_t* awq_qweight, const uint32_t* awq_qzeros, const void* awq_scales, int K, int N,
                    int group_size, void* weight_kn_out, int8_t* zeros_out, void* stream) {
    if (!awq_qweight || !awq_qzeros || !awq_scales || !weight_kn_out || !zeros_out)
        return fail(GPTQ_ERR_NULL, "awq_qweight/awq_qzeros/awq_scales/weight_kn_out/zeros_out must be non-NULL");
    if (int rc = awq_shape_check(K, N, group_size)) return rc;
    hipError_t e = launch_awq_unpack(awq_qweight, awq_qzeros, awq_scales, K, N, group_size, weight_kn_out, zeros_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_awq_unpack launch");
    return GPTQ_OK;
}

int gptq_awq_repack(const uint32_t* awq_qweight, const uint32_t* awq_qzeros, int K, int N, int group_size,
                    uint32_t* qweight_out, uint32_t* qzeros_out, void* stream) {
    if (!awq_qweight || !awq_qzeros || !qweight_out || !qzeros_out)
        return fail(GPTQ_ERR_NULL, "awq_qweight/awq_qzeros/qweight_out/qzeros_out must be non-NULL");
    if (int rc = awq_shape_check(K, N, group_size)) return rc;
    hipError_t e = launch_awq_repack(awq_qweight, awq_qzeros, K, N, group_size, qweight_out, qzeros_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_awq_repack launch");
    return GPTQ_OK;
}


// ---- direct peer-store all-gather (peer.hip) ---------------------------------------------------------------------------
static int check_peer_group(const gptq_peer_group_t* pg, int M, int dtype) {
    if (!pg) return fail(GPTQ_ERR_NULL, "peer group is NULL");
    if (pg->world < 1 || pg->world > GPTQ_PEER_MAX) return fail(GPTQ_ERR_SHAPE, "peer group world (%d) must be 1..%d", pg->world, GPTQ_PEER_MAX);
    if (pg->rank < 0 || pg->rank >= pg->world) return fail(GPTQ_ERR_SHAPE, "peer group rank (%d) outside 0..%d", pg->rank, pg->world - 1);
    if (!pg->state) return fail(GPTQ_ERR_NULL, "peer group state is NULL");
    for (int r = 0; r < pg->world; ++r)
        if (!pg->xbuf[0][r] || !pg->xbuf[1][r] || !pg->flags[r]) return fail(GPTQ_ERR_NULL, "peer group: buffers / flags of rank %d are NULL", r);
    if (!dtype_ok(dtype)) return fail(GPTQ_ERR_UNSUPPORTED, "unsupported dtype enum %d", dtype);
    if (pg->N <= 0 || pg->N % pg->world || (pg->N / pg->world) % 32)
        return fail(GPTQ_ERR_SHAPE, "peer group: out_features (%d) must split over %d ranks in multiples of 32 columns", pg->N, pg->world);
    if (M <= 0 || M > pg->rows_max) return fail(GPTQ_ERR_SHAPE, "M (%d) must be 1..rows_max (%d) of the exchange buffers", M, pg->rows_max);
    return GPTQ_OK;
}

int gptq_peer_scatter(const gptq_peer_group_t* pg, const void* y_local, int M, int n_local, int dtype, void* stream) {
    if (int rc = check_peer_group(pg, M, dtype)) return rc;
    if (!y_local) return fail(GPTQ_ERR_NULL, "y_local must be non-NULL");
    if (n_local != pg->N / pg->world) return fail(GPTQ_ERR_SHAPE, "n_local (%d) must be N / world = %d", n_local, pg->N / pg->world);
    hipError_t e = launch_peer_scatter(*pg, y_local, M, n_local, dtype, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_peer_scatter launch");
    return GPTQ_OK;
}

int gptq_peer_publish(const gptq_peer_group_t* pg, void* stream) {
    if (int rc = check_peer_group(pg, 1, GPTQ_F16)) return rc;
    hipError_t e = launch_peer_publish(*pg, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_peer_publish launch");
    return GPTQ_OK;
}

int gptq_peer_collect(const gptq_peer_group_t* pg, void* out, int M, int dtype, uint32_t max_spins, void* stream) {
    if (int rc = check_peer_group(pg, M, dtype)) return rc;
    if (!out) return fail(GPTQ_ERR_NULL, "out must be non-NULL");
    if (max_spins == 0) return fail(GPTQ_ERR_SHAPE, "max_spins must be > 0 (the wait is bounded by design)");
    hipError_t e = launch_peer_collect(*pg, out, M, dtype, max_spins, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "gptq_peer_collect launch");
    return GPTQ_OK;
}

// The column shard's decode kernel with the scatter as its epilogue: local kernel + one collect launch per tensor-parallel layer.
int gptq_forward_scatter(const gptq_layer_t* L, const void* x, int M, const gptq_peer_group_t* pg, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_layer(L);
    if (rc) return rc;
    if (!x) return fail(GPTQ_ERR_NULL, "x must be non-NULL");
    if ((rc = check_peer_group(pg, M, L->dtype))) return rc;
    if (L->N != pg->N / pg->world) return fail(GPTQ_ERR_SHAPE, "the layer's out_features (%d) must be the rank's shard N / world = %d", L->N, pg->N / pg->world);
    const gptq_layer_t* one[1] = {L};
    if (M > 4 || L->epilogue != GPTQ_EPI_NONE || L->g_idx || !want_tiled(one, 1, M, nullptr) || (plan_tiled(one, 1, M, nullptr).u != 2 && plan_tiled(one, 1, M, nullptr).u != 4))
        return fail(GPTQ_ERR_UNSUPPORTED, "gptq_forward_scatter: the fused scatter is the epilogue of the decode-copy kernel (M <= 4, a plain 3/4/8-bit fp16/bf16 "
                                          "layer that carries qweight_tiled / qconst_tiled); use gptq_forward + gptq_peer_scatter for this call");
    const WsView wv = split_ws(ws, ws_bytes);
    const TiledPlan tp = plan_tiled(one, 1, M, nullptr);
    if (tp.partial_bytes > 0 && wv.body_bytes < tp.partial_bytes)
        return fail(GPTQ_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", WS_HEADER_BYTES + tp.partial_bytes, wv.header ? WS_HEADER_BYTES + wv.body_bytes : (size_t)0);
    void* outs[1] = {nullptr};                                  // the rank's own slice travels through its own exchange buffer like every peer's
    hipError_t e = launch_tiled(one, tp, x, outs, M, wv.header, wv.body, (hipStream_t)stream, pg);
    if (e != hipSuccess) return hip_fail(e, "gptq_forward_scatter launch (was gptq_init() called on this device?)");
    return GPTQ_OK;
}

int gptq_forward_gather(const gptq_layer_t* L, const void* x, void* out, int M, const gptq_peer_group_t* pg, uint32_t max_spins, void* ws, size_t ws_bytes,
                        void* stream) {
    if (!out) return fail(GPTQ_ERR_NULL, "out must be non-NULL");
    if (max_spins == 0) return fail(GPTQ_ERR_SHAPE, "max_spins must be > 0 (the wait is bounded by design)");
    if (int rc = gptq_forward_scatter(L, x, M, pg, ws, ws_bytes, stream)) return rc;
    return gptq_peer_collect(pg, out, M, L->dtype, max_spins, stream);
}

int gptq_peer_gather(const gptq_peer_group_t* pg, const void* y_local, void* out, int M, int n_local, int dtype,
                     uint32_t max_spins, void* stream) {
    if (!out) return fail(GPTQ_ERR_NULL, "out must be non-NULL");
    if (max_spins == 0) return fail(GPTQ_ERR_SHAPE, "max_spins must be > 0 (the wait is bounded by design)");
    if (int rc = gptq_peer_scatter(pg, y_local, M, n_local, dtype, stream)) return rc;
    return gptq_peer_collect(pg, out, M, dtype, max_spins, stream);
}

}  // extern "C"
