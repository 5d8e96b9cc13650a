// launch.h -- host-side declarations shared between the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/gptq_mi355x.h"

namespace gptq {

struct GemvPlan {
    int ln, waves, ksplit, strips, mt, mtiles;
    int units_total, units_per_split, chunk_units;
    bool perk, use_seq;
    bool mfma;     // 4-bit fp16 / bf16 register kernel with the k-reduction on v_mfma_f32_4x4x4_16b_f16
    bool mfmag;    // matrix-core kernel for the other packings / bf16 (gemv_mfma_generic_kernel)
    bool magic;    // ... with the packed magic-number field decode (3- / 8-bit fp16)
    bool pair;     // mfma path with the fused SILU_MUL epilogue (gate/up halves walked by the same workgroup)
    bool xperm;    // act-order, 2+ rows of x: x is permuted once by a pre-pass into the workspace front (xperm_bytes) and the plain kernel streams qweight_seq
    size_t xperm_bytes;
    int u;         // direct path: consecutive packed rows per lane and iteration
    size_t lds_bytes, workspace_bytes;
};
GemvPlan plan_gemv(const gptq_layer_t& L, int M, const gptq_tuning_t* tune);
hipError_t launch_gemv(const gptq_layer_t& L, const GemvPlan& pl, const void* x, void* out, int M,
                       void* workspace, hipStream_t st);

// 17 .. 128 rows (gemm_mid.hip: gemm_mid_kernel): 1..4 plain 4-bit layers that read the same x, one launch; weights, x and group constants by LDS DMA.
struct MidPlan {
    bool ok;                 // every layer qualifies and the geometry fits
    bool pays;               // measured preference over the other kernels
    bool xreg;               // experiment: x through registers (ordinary loads + ds_write) instead of LDS DMA
    bool gran;               // K slices combined through tag-validated {fp32, tag} granules (one hop) instead of partial tiles + flags
    int cw;                  // 64-column halves per strip (2: 128-column strips, M <= 64)
    int row_blocks;          // workgroups along M (each owns 16 rt rows of x)
    int bits;                // 4 or 8
    int nseg, rt, waves, stages, ksplit, ksteps_total, ksteps_per_split, strips_total, nsum, lg_gsteps, tab_bytes;
    size_t lds_bytes;
    size_t partial_bytes;    // behind the header (and the permuted x of an act-order layer): [ksplit - 1][M][nsum] fp32 when ksplit > 1
};
MidPlan plan_mid(const gptq_layer_t* const* layers, int n, int M, const gptq_tuning_t* tune);
// qweight_override: the re-sequenced rows of a single act-order layer (x is then the permuted copy), else NULL
hipError_t launch_mid(const gptq_layer_t* const* layers, const MidPlan& pl, const void* x, void* const* outs, int M, void* ws_header, void* partial,
                      const uint32_t* qweight_override, hipStream_t st);
hipError_t init_gemm_mid_device();

// gemm_wide_sk.hip: the same wave tile as a stream-K partition (one persistent workgroup per CU; 128 x 256 tiles, two K parts per workgroup) for launches
// that do not divide into whole rounds of 128 x 512 tiles; weights from the decode copy only
struct WideSkGeom {
    int nbm, nbn, upt, units_total, lg_nwg;
    size_t slot_bytes;        // published accumulators of cut tiles: 256 KiB per workgroup (0 when every range boundary is a tile boundary)
};

// gemm_rows.hip: 5 .. ~256 rows from the decode copy, a workgroup = 16 rb rows x s strips x the whole K, no exchange between workgroups
struct RowsPlan {
    bool ok;
    int rb, s, xbufs, waves, cpw, npm, nsg;
    size_t lds_bytes;
};
bool rows_ok(const gptq_layer_t& L, int M);
bool rows_pays(const gptq_layer_t& L, int M);
RowsPlan plan_rows(const gptq_layer_t& L, int M, const gptq_tuning_t* tune);
hipError_t launch_gemm_rows(const gptq_layer_t& L, const RowsPlan& pl, const void* x, void* out, int M, hipStream_t st);
bool rows_multi_ok(const gptq_layer_t* const* Ls, int n, int M);      // 1 .. 4 layers that read the same x in ONE launch (gptq_forward_multi)
bool rows_multi_pays(const gptq_layer_t* const* Ls, int n, int M);
RowsPlan plan_rows_multi(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* tune);
hipError_t launch_gemm_rows_multi(const gptq_layer_t* const* Ls, int n, const RowsPlan& pl, const void* x, void* const* outs, int M, hipStream_t st);
hipError_t init_gemm_rows_device();

// gemm_panel.hip: 129 .. ~767 rows from the decode copy, a workgroup = 32 mt rows x 32 nt columns x the whole K (kp waves = K parts), no exchange between workgroups
struct PanelPlan {
    bool ok;
    int mt, nt, kp, nbm, nbn, spw;
    size_t lds_bytes;
};
bool panel_ok(const gptq_layer_t& L, int M);
bool panel_pays(const gptq_layer_t& L, int M);
bool panel_pays_filled(const gptq_layer_t& L, int M);
PanelPlan plan_panel(const gptq_layer_t& L, int M, const gptq_tuning_t* tune);
hipError_t launch_gemm_panel(const gptq_layer_t& L, const PanelPlan& pl, const void* x, void* out, int M, hipStream_t st);
hipError_t init_gemm_panel_device();

struct GemmPlan {
    bool supported, use_seq;
    bool panel;               // gemm_panel.hip (panelp holds its geometry); act-order layers: x permuted in natural order (xnat)
    PanelPlan panelp;
    bool rows;                // gemm_rows.hip (rowsp holds its geometry); act-order layers: x permuted in natural order (xnat)
    RowsPlan rowsp;
    bool wsk;                 // stream-K partition of 128 x 256 tiles with the 128 x 128 wave tile (gemm_wide_sk.hip); implies wide_tiled (+ xnat for act-order layers)
    WideSkGeom wskg;
    bool wide_tiled;          // ... reading the layer's decode copy, raw x staged by LDS DMA
    bool xnat;                // act-order + wide_tiled: x permuted in natural order (the copy is of the re-sequenced rows)
    bool wide;                // 128 x 512 tiles, 128 x 128 per wave, accumulators in AGPRs (gemm_wide.hip): large launches
    bool mid;                 // 17 .. 128 rows, 4-bit: gemm_mid_kernel (midp holds its geometry)
    MidPlan midp;
    bool glds;                // ... and stages it with global_load_lds (DMA) into swizzled, unpadded LDS rows
    bool xslot;               // act-order: the x pre-pass writes k-slot order, the kernel copies x to LDS verbatim
    bool strip16;             // 8 < M <= 64, 4-bit: 16-column strips on v_mfma_f32_16x16x32 (mt = row tiles of 16)
    bool f32;                 // fp32 I/O: exact-f32 matrix core kernel (128 x 128 tiles)
    bool skinny;              // weight-streaming decomposition for 8 < M <= 128 (64-column strips, waves split K)
    bool stream64;            // batched decode 4 < M <= 64, 4-bit: 64-column strips by LDS DMA, in-launch K-split combine (u = K-steps in flight per wave)
    int waves, variant, u;
    int kg;                   // K groups inside a workgroup (2 = 8 waves, two K halves summed through LDS)
    int mt, bk, bm, bn;       // row tiles per wave, K-step, workgroup tile
    int nbm, nbn, ksplit, ksteps_total, ksteps_per_split;
    int tail, tail_lg;        // tiled kernel, balanced tail: the last `tail` tiles run as 2^tail_lg K slices (one workgroup each, combined inside the launch)
    size_t xperm_bytes;       // permuted-x scratch for act-order layers (front of the workspace)
    size_t workspace_bytes;   // xperm + split-K partial slabs (or the balanced tail's accumulator slabs)
};
GemmPlan plan_gemm(const gptq_layer_t& L, int M, const gptq_tuning_t* tune);
hipError_t launch_gemm(const gptq_layer_t& L, const GemmPlan& pl, const void* x, void* out, int M,
                       void* ws_header, void* workspace, hipStream_t st, bool x_permuted = false);

// Streamed GEMV (gemv_q4_stream_kernel): 1..4 plain 4-bit layers that read the same x, one launch.
constexpr size_t WS_HEADER_BYTES = 65536;      // front of every workspace: arrival tickets of the in-launch K-split combine (kept zero)
constexpr size_t WS_HEADER_EPOCH_OFFSET = 32768; // second half of the header: per-strip launch epochs of the streamed GEMV's one-hop K-split combine (monotonic, never reset)
constexpr size_t WS_HEADER_TAIL_BYTES = 64;    // ... except its last 64 bytes: launch epoch / arrival count / sticky error word of the fused MLP exchange (mlp.hip)
struct StreamPlan {
    bool ok;                 // every layer qualifies and the geometry fits
    int nseg, ln, waves, u, mt, ksplit, units_total, units_per_split, strips_total, nsum;
    size_t lds_bytes;
    size_t partial_bytes;    // behind the header: [ksplit][M][nsum] fp32 when ksplit > 1
};
StreamPlan plan_stream(const gptq_layer_t* const* layers, int n, int M, const gptq_tuning_t* tune);
hipError_t launch_stream(const gptq_layer_t* const* layers, const StreamPlan& pl, const void* x, void* const* outs, int M,
                         void* ws_header, void* ws_body, hipStream_t st);
// Batched decode (gemm_stream64_kernel): 1..4 plain 4-bit layers that read the same x, 1 <= M <= 64, one launch.
struct Stream64Plan {
    bool ok;                 // every layer qualifies and the geometry fits
    bool pays;               // measured preference over the per-layer kernels (enough workgroups, not a small layer)
    int nseg, mt, waves, u, ksplit, ksteps_total, ksteps_per_split, strips_total, nsum;
    size_t lds_bytes;
    size_t partial_bytes;    // behind the header (and the permuted x of an act-order layer): [ksplit][M][nsum] fp32 when ksplit > 1
};
Stream64Plan plan_stream64(const gptq_layer_t* const* layers, int n, int M, const gptq_tuning_t* tune);
// qweight_override: the re-sequenced rows of a single act-order layer (x is then the permuted copy), else NULL
hipError_t launch_stream64(const gptq_layer_t* const* layers, const Stream64Plan& pl, const void* x, void* const* outs, int M,
                           void* ws_header, void* partial, const uint32_t* qweight_override, hipStream_t st);
// Decode from the load-time decode copy (gemv_tiled.hip: gemv_tiled_kernel): 1..4 plain 3/4/8-bit layers with qweight_tiled / qconst_tiled that read the same x, M <= 4, one launch.
struct TiledPlan {
    bool ok;                 // every layer qualifies and the geometry fits
    int nstr;                // strips per workgroup (1; 2 / 4: gemv_tiled_multi.hip -- adjacent strips of a layer behind one staged x)
    bool pair;               // one [gate | up] layer with the SILU_MUL epilogue: a workgroup per pair of strips, SiLU * mul behind the cross-wave sum
    int nseg, bits, waves, u, mt, ksplit, chunks_total, chunks_per_split, strips_total, nsum, groups, xstride, xraw_off;
    size_t lds_bytes;
    size_t partial_bytes;    // behind the header: [ksplit - 1][M][nsum] granules when ksplit > 1
    bool zm2;                // bf16 4-bit layers at 2 rows, launches below 1024 workgroups: the zero-point on the matrix core (gemv_tiled_kernel<..., ZM2 = 1>)
};
bool tiled_layer_ok(const gptq_layer_t& L);
TiledPlan plan_tiled(const gptq_layer_t* const* layers, int n, int M, const gptq_tuning_t* tune);
hipError_t launch_tiled(const gptq_layer_t* const* layers, const TiledPlan& pl, const void* x, void* const* outs, int M, void* ws_header, void* ws_body,
                        hipStream_t st, const gptq_peer_group_t* pg = nullptr);      // pg: tensor-parallel epilogue (outs[0] may be null then)
hipError_t init_gemv_tiled_device();
// the decode copy of a 3/4/8-bit layer (utils.hip): qweight_tiled (chunks of 4 k-slots x 16 columns, column per lane, fields in pair order) and qconst_tiled
size_t tiled_weight_bytes(const gptq_layer_t& L);
size_t tiled_const_bytes(const gptq_layer_t& L);
hipError_t launch_prepack_decode(const uint32_t* qweight, const uint32_t* qzeros, const void* scales, int K, int N, int bits, int group_size, int zero_mode,
                                 uint32_t* tiled_out, void* const_out, hipStream_t st);
hipError_t launch_unprepack_decode(const uint32_t* tiled, int K, int N, int bits, uint32_t* qweight_out, hipStream_t st);      // the exact inverse (weights)
bool stream_preferred(const gptq_layer_t& L, int M);                              // single layer: streamed kernel instead of the register one?
bool multi_preferred(const gptq_layer_t* const* layers, int n, int M);             // several layers sharing x: one streamed launch?
hipError_t launch_silu_mul2(const void* g, const void* u, void* out, size_t total, int dtype, hipStream_t st);
bool silu_mul2_permute_ok(int K, int dtype);            // mlp.hip: SiLU * mul and the x permute of an act-order down projection in one pass
hipError_t launch_silu_mul2_permute(const void* g, const void* u, const int32_t* perm, int M, int K, int dtype, void* out, hipStream_t st);
hipError_t init_mlp_device();
// gemm_wide.hip: 128 x 512 prefill tiles, 128 x 128 per wave with the accumulators in AGPRs (4-bit fp16 / bf16; glds: x in k-slot order, staged by LDS DMA)
bool wide_gemm_ok(const gptq_layer_t& L, int M, bool use_seq, bool xslot_glds);
hipError_t launch_gemm_wide(const gptq_layer_t& L, const uint32_t* qweight, const void* x, void* out, int M, bool glds, hipStream_t st, bool tiled = false);
bool wide_sk_ok(const gptq_layer_t& L, int M);
bool wide_sk_pays(const gptq_layer_t& L, int M);      // the planner's measured preference over the whole-tile kernels
WideSkGeom wide_sk_geom(const gptq_layer_t& L, int M);
hipError_t launch_gemm_wide_sk(const gptq_layer_t& L, const void* x, void* out, int M, void* ws_header, void* slots, hipStream_t st);
hipError_t init_gemm_wide_sk_device();
hipError_t init_gemv_device();
hipError_t init_gemm_device();

hipError_t launch_dequant(const gptq_layer_t& L, void* W_out, hipStream_t st);
hipError_t launch_unpack_weights(const uint32_t* qweight, int K, int N, int bits, uint8_t* w_out, hipStream_t st);
hipError_t launch_unpack_zeros(const uint32_t* qzeros, int G, int N, int bits, int zero_mode, int32_t* z_out, hipStream_t st);
hipError_t launch_pack_weights(const void* W, const void* scale_in, const void* zero_in, const int32_t* g_idx,
                               int K, int N, int bits, int group_size, int w_dtype, int qparam_dtype,
                               uint32_t* qweight_out, void* scales_out, hipStream_t st);
hipError_t launch_pack_zeros(const void* zero_in, int G, int N, int bits, int qparam_dtype, uint32_t* qzeros_out, hipStream_t st);
hipError_t launch_resequence(const uint32_t* qweight, const int32_t* perm, int K, int N, int bits,
                             uint32_t* out, hipStream_t st);
hipError_t launch_awq_unpack(const uint32_t* aq, const uint32_t* az, const void* scales, int K, int N, int group_size, void* w_kn, int8_t* zeros, hipStream_t st);
hipError_t launch_awq_repack(const uint32_t* aq, const uint32_t* az, int K, int N, int group_size, uint32_t* qweight, uint32_t* qzeros, hipStream_t st);
hipError_t launch_silu_mul(const void* y, void* out, int M, int N, int dtype, hipStream_t st);
hipError_t launch_permute_rows16(const void* x, const int32_t* perm, int M, int K, void* x_out, hipStream_t st, bool slot_order = false);
hipError_t launch_permute_columns(const void* x, const int32_t* perm, int M, int K, int dtype, void* x_out, hipStream_t st);
// peer.hip: direct peer-store all-gather (y_local / out may be NULL: scatter-only / collect-only)
hipError_t launch_peer_scatter(const gptq_peer_group_t& pg, const void* y_local, int M, int n_local, int dtype, hipStream_t st);
hipError_t launch_peer_publish(const gptq_peer_group_t& pg, hipStream_t st);
hipError_t launch_peer_collect(const gptq_peer_group_t& pg, void* out, int M, int dtype, unsigned max_spins, hipStream_t st);

}  // namespace gptq
