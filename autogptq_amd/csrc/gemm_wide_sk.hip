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
// (the two K parts already summed: write-through stores, 32 KiB per wave) and raises a flag without ever waiting for anybody, so a finisher only waits for workgroups that have nothing
// in front of their publish (no residency assumption beyond "every workgroup is eventually scheduled"; the grid is one workgroup per CU anyway).  Sums
// are formed in a fixed order (part 0 + part 1 of every piece through LDS, then the published pieces in range order): bit-reproducible.
// The last row tile is SHIFTED UP to end at row M (m0 = M - 128) instead of clamping rows: no x address of the DMA ever needs a per-lane clamp (M >= 128);
// the overlapped rows are computed twice and stored once, by the tile they belong to.
// Weights: the layer's decode copy (gptq_prepack_decode), raw x by LDS DMA, the exact magic-number dequant -- the loop body is gemm_wide_kernel<T, true,
// true, G128>'s (gemm_wide.hip), minus its ablation switches.
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"
#include "gemm_wide_sk_kernel.cuh"
#include "launch.h"

namespace gptq {
namespace wide {

template <typename T, bool G128>
__global__ void __launch_bounds__(256, 1) gemm_wide_sk_kernel(WskParams p) {
    wsk_body<T, 4, G128 ? 0 : 1>(p);
}

}  // namespace wide

// ---- host side -------------------------------------------------------------------------------------------------------------------------------
// group mode of the kernel body (gemm_wide_sk_kernel.cuh): 0 = 128-multiples, 1 = 64-multiples, 2 = 32-wide groups; -1 = not served
int wide_sk_group_mode(int group_size) { return group_size % 128 == 0 ? 0 : (group_size % 64 == 0 ? 1 : (group_size == 32 ? 2 : -1)); }

bool wide_sk_ok(const gptq_layer_t& L, int M) {
    if ((L.bits != 4 && L.bits != 3 && L.bits != 8) || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16)) return false;
    if (L.qweight_tiled == nullptr || L.tiled_cols != GPTQ_STRIP_COLS) return false;
    if (L.bits != 4 && L.qconst_tiled == nullptr) return false;                                      // 3 / 8 bits read the copy's constant records
    if (L.K % 256 || wide_sk_group_mode(L.group_size) < 0 || L.N % 32 || L.epilogue != GPTQ_EPI_NONE) return false;      // two K parts of whole 128-deep chunks
    return M >= 128;
}

// Measured preference over the whole-tile kernels (tools/wide_sk_ab.py, profiles/r05_wide_sk_sweep.log: layer call incl. the x permute of act-order layers, us,
// 128 x 256 tiles -> this, M = 2048: 4096^2 69.2 -> 63.5 (act-order 75.4 -> 71.2), 4096x11008 202.9 -> 178.7 (200.5 -> 191.0), 11008x4096 175.6 -> 163.3
// (191.4 -> 182.6); 1.03 - 1.28x from 768 rows up on every shape, 0.87 - 0.89x at 512 rows on 4096^2 (each tile cut in four) and 1.06 - 1.12x there on the
// two large shapes).  Where whole 128 x 512 tiles fill their rounds (M = 4096 / 8192 on these shapes) gemm_wide_kernel stays: 0.90 - 1.00x -- the
// caller (plan_gemm) asks that first.
bool wide_sk_pays(const gptq_layer_t& L, int M) {
    if (!wide_sk_ok(L, M)) return false;
    // 3 / 8 bits and 32-wide groups have no other 64-deep kernel (the row kernel runs them in BK = 32 steps): 1.13 - 1.92x (8 bits: 1.02 - 1.50x) from 512 rows on every shape
    // (profiles/r05_wide_sk_b38_ab.log: int3 g32 M = 2048: 109 -> 64, 257 -> 188, 276 -> 163 us; int4 g32: 92 -> 63, 211 -> 189, 229 -> 164 us)
    // below: 384 rows 1.18 - 2.09x on the two large shapes (0.97 - 1.05x on 4096^2), 256 rows 1.24 - 1.51x on 4096 -> 11008 only (0.71 - 0.94x elsewhere)
    if (L.bits != 4 || wide_sk_group_mode(L.group_size) == 2)
        return M >= 512 || (M >= 384 && (size_t)L.K * L.N >= ((size_t)32 << 20)) || (M >= 256 && L.N >= 2 * L.K && (size_t)L.K * L.N >= ((size_t)32 << 20));
    // late round 6 (tools/mid_band_sweep.py over the larger families, profiles/r06_mid_band_sweep4.log, default -> this kernel at 384 rows): wide / square layers of 56+ Mi
    // weights 5120x13824 72.6 -> 62.3 us, 8192^2 71.3 -> 61.6, 4096x14336 61.4 -> 54.0, 6656x17920 116.3 -> 95.8; the deep ones lose there (13824x5120 58.5 against 70.1,
    // 14336x4096 54.7 / 62.9) except the largest from 320 rows (17920x6656 114.6 -> 101.5, 28672x8192 190.9 -> 173.2)
    const size_t kn = (size_t)L.K * L.N;
    if (M >= 384 && kn >= ((size_t)56 << 20) && L.N >= L.K) return true;
    if (M >= 320 && kn >= ((size_t)112 << 20) && L.K > L.N) return true;
    return M >= 768 || (M >= 512 && kn >= ((size_t)32 << 20));
}

namespace mlpk { extern int g_cu_count[64]; }                 // per device ordinal, filled by gptq_init (mlp.hip); 0 = not initialised

// One persistent workgroup per CU: the grid is the largest power of two <= the CU count of the calling thread's current device (256 on an MI355X; a partitioned
// or smaller device gets a smaller grid, so that a finisher's bounded wait never depends on publisher workgroups that are not resident yet).  Before gptq_init()
// has recorded the count (host-only plan queries): 256.
static int wide_sk_lg_nwg_cap() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) cus = mlpk::g_cu_count[dev];
    if (cus <= 0 || cus > 256) cus = 256;
    int lg = 0;
    while ((2 << lg) <= cus) ++lg;
    return lg;
}

WideSkGeom wide_sk_geom(const gptq_layer_t& L, int M) {
    WideSkGeom g{};
    g.nbm = (M + 127) / 128;
    g.nbn = (L.N + 255) / 256;
    g.upt = L.K / 256;
    const long units = (long)g.nbm * g.nbn * g.upt;
    g.units_total = (int)units;
    g.lg_nwg = wide_sk_lg_nwg_cap();                          // one workgroup per CU
    while (g.lg_nwg > 0 && (1L << g.lg_nwg) > units) --g.lg_nwg;
    // does any range boundary fall inside a tile?  (then pieces are published: 128 KiB per workgroup behind the permuted x)
    bool cut = false;
    for (int b = 1; b < (1 << g.lg_nwg) && !cut; ++b) cut = (((long)b * units) >> g.lg_nwg) % g.upt != 0;
    g.slot_bytes = cut ? ((size_t)1 << g.lg_nwg) * wide::WSK_SLOT_FLOATS * sizeof(float) : 0;
    return g;
}

void launch_gemm_wide_skb(int bits, int gm, int dtype, dim3 grid, dim3 block, hipStream_t st, const wide::WskParams& p);      // gemm_wide_sk_b38.hip
hipError_t init_gemm_wide_skb_device();

template <typename T, bool G128>
static hipError_t grant_wsk() {
    return hipFuncSetAttribute((const void*)wide::gemm_wide_sk_kernel<T, G128>, hipFuncAttributeMaxDynamicSharedMemorySize, wide::WSK_LDS_BYTES);
}
template <typename T, bool G128>
static void launch_wsk(dim3 grid, dim3 block, hipStream_t st, const wide::WskParams& p) {
    hipLaunchKernelGGL((wide::gemm_wide_sk_kernel<T, G128>), grid, block, wide::WSK_LDS_BYTES, st, p);
}
hipError_t init_gemm_wide_sk_device() {
    hipError_t e = grant_wsk<f16, true>();
    if (e == hipSuccess) e = grant_wsk<f16, false>();
    if (e == hipSuccess) e = grant_wsk<bf16, true>();
    if (e == hipSuccess) e = grant_wsk<bf16, false>();
    if (e == hipSuccess) e = init_gemm_wide_skb_device();
    return e;
}

hipError_t launch_gemm_wide_sk(const gptq_layer_t& L, const void* x, void* out, int M, void* ws_header, void* slots, hipStream_t st) {
    if (!wide_sk_ok(L, M)) return hipErrorInvalidValue;
    const WideSkGeom g = wide_sk_geom(L, M);
    if (g.slot_bytes && (!ws_header || !slots)) return hipErrorInvalidValue;
    wide::WskParams p{};
    p.qweight = L.qweight_tiled; p.qconst = L.qconst_tiled; p.qzeros = L.qzeros; p.scales = L.scales; p.bias = L.bias; p.x = x; p.out = out;
    p.M = M; p.K = L.K; p.N = L.N; p.zero_mode = L.zero_mode;
    p.nbm = g.nbm; p.nbn = g.nbn; p.upt = g.upt; p.units_total = g.units_total; p.lg_nwg = g.lg_nwg;
    p.chunks = L.bits == 8 ? L.K / 64 : L.K / 128;          // the decode copy's chunks: 4 k-slots of 32 (8 bits: 16) values
    p.groups = (L.K + L.group_size - 1) / L.group_size;
    const unsigned long long kpg = (unsigned long long)(L.group_size >= 64 ? L.group_size / 64 : 1);
    p.kpg_inv = ((1ull << 32) + kpg - 1) / kpg;
    p.max_spins = 1u << 22;
    p.flags = (unsigned*)ws_header;
    p.slots = (float*)slots;
    p.err = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2 : nullptr;
    const dim3 grid(1 << g.lg_nwg), block(256);
    const int gm = wide_sk_group_mode(L.group_size);
    if (L.bits != 4 || gm == 2) {                              // gemm_wide_sk_b38.hip
        launch_gemm_wide_skb(L.bits, gm, L.dtype, grid, block, st, p);
        return hipGetLastError();
    }
    const bool g128 = gm == 0;
    if (L.dtype == GPTQ_F16) {
        if (g128) launch_wsk<f16, true>(grid, block, st, p);
        else launch_wsk<f16, false>(grid, block, st, p);
    } else {
        if (g128) launch_wsk<bf16, true>(grid, block, st, p);
        else launch_wsk<bf16, false>(grid, block, st, p);
    }
    return hipGetLastError();
}

}  // namespace gptq
