/* gptq_mi355x.h -- C ABI of libgptq_mi355x.so: the MI355X (gfx950) quantized-linear hot path.
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers, ints and an
 * opaque stream handle (a hipStream_t passed as void*); no torch / pybind types.  Each one
 * replaces a pybind11 function (or group of functions) of the reference's native extensions,
 * cited as `file:line` into the AutoGPTQ tree (v0.8.0.dev0):
 *
 *   gptq_forward / gptq_gemv / gptq_gemm
 *       <- autogptq_cuda_{64,256}: vecquant{2,3,4,8}matmul            autogptq_extension/cuda_256/autogptq_cuda_256.cpp:5-64,175-178
 *                                  vecquant{2,3,4,8}matmul_old        :68-126,180-183
 *                                  vecquant{2,3,4}matmul_faster_old   :128-171,184-186
 *       <- exllama_kernels.q4_matmul                                  autogptq_extension/exllama/exllama_ext.cpp:176-217,258
 *       <- exllamav2_kernels.gemm_half_q_half                         autogptq_extension/exllamav2/ext.cpp:95-126,133
 *       <- autogptq_marlin_cuda.mul                                   autogptq_extension/marlin/marlin_cuda.cpp:30-75,78
 *       and the pure-PyTorch fallback they all share                  auto_gptq/nn_modules/qlinear/qlinear_cuda_old.py:291-355, qlinear_cuda.py:253-317
 *   gptq_dequant
 *       <- the `reconstruct` step of exllama / exllamav2              exllama/cuda_func/q4_matrix.cu:171-225, exllamav2/cuda/q_matrix.cu:158-279,452-500
 *   gptq_make_sequential + gptq_resequence_qweight + gptq_permute_columns
 *       <- exllama_kernels.make_q4 (Q4Matrix::make_sequential)        exllama/exllama_ext.cpp:134-171, cuda_func/q4_matrix.cu:63-169
 *          exllamav2_kernels.make_q_matrix                            exllamav2/ext.cpp:26-93, cuda/q_matrix.cu:502-627
 *          column_remap_cuda                                          exllama/cuda_func/column_remap.cu:9-63
 *       (unlike those, never mutates qweight in place: results go to caller-owned side buffers)
 *   gptq_prepack_decode / gptq_prepack_decode_bytes
 *       <- the load-time weight re-layouts: exllamav2 shuffle_kernel   exllamav2/cuda/q_matrix.cu:19-42,149
 *          exllama Q4Matrix::make_sequential's row rewrite             exllama/cuda_func/q4_matrix.cu:105-169
 *          Marlin's repack kernel                                      marlin/marlin_repack.cu:8-92
 *       (into a caller-owned side buffer; the checkpoint tensor stays as it is)
 *   gptq_pack_weights / gptq_pack_zeros
 *       <- QuantLinear.pack (CPU-only in the reference)               auto_gptq/nn_modules/qlinear/qlinear_cuda.py:108-203
 *   gptq_unpack_weights / gptq_unpack_zeros
 *       <- the integer unpack of the Python path                      qlinear_cuda_old.py:295-344, qlinear_cuda.py:257-295
 *
 * Tensor layout = the GPTQ v1 checkpoint ABI (K = in_features, N = out_features):
 *   qweight uint32 [K/32*bits, N]   row-major; values packed along K, LSB first
 *   qzeros  uint32 [G, N/32*bits]   G = ceil(K/group_size); packed along N; stores zero-1
 *   scales  dtype  [G, N]
 *   g_idx   int32  [K] or NULL      NULL = sequential groups (k / group_size)
 *   bias    dtype  [N] or NULL
 *   x       dtype  [M, K] row-major; out dtype [M, N] row-major
 *
 * Conventions
 *   - every function returns GPTQ_OK (0) or a gptq_status_t > 0; gptq_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - nothing allocates or frees device memory; scratch is caller-provided and sized by
 *     gptq_workspace_bytes().  The first GPTQ_WORKSPACE_HEADER_BYTES of a workspace hold the arrival
 *     tickets of the in-launch K-split combine: they must be ZERO when the buffer is first handed to
 *     the library (one hipMemset at allocation).  Its first half (32 KiB: arrival tickets, and the per-slice
 *     flag words of the 17..128-row kernel's K-split combine, which its owner slices clear again) is left zero by
 *     every launch; its second half carries state from one launch to the next and is NOT zero afterwards: one
 *     launch-epoch word per column strip for the streamed GEMV's K-split combine (partial sums travel as
 *     {value, tag} granules through the body and are validated by their tag), and in the last 64 bytes the
 *     launch epoch / arrival count / sticky error word ([2]: a bounded wait gave up) of the exchanges.
 *     One workspace serves one stream at a time (launches that may overlap need their own).  Kernels are enqueued on the caller's stream, never synchronise,
 *     and are legal inside hipGraph capture: besides the kernel launches the forward entry points make one kind of runtime-API call, hipGetDevice(), and only
 *     for launches whose workgroups wait for each other inside the launch (K slices of the 17..256-row kernel): their grid is checked against the CU count.
 *     Process-global state of the library: the per-device CU count and the > 64 KiB dynamic-LDS grants, both written once per device by gptq_init(), which
 *     the caller runs once per device, outside any capture, before the first forward (a launch that needs the CU count and does not find it is refused).
 *     A bounded wait inside a launch that gives up (another process held the CUs) sets word [2] of the header's last 64 bytes and never hangs the queue:
 *     that launch's result is wrong and the word stays set -- poll it (autogptq_amd.qlinear_mi355x.exchange_error) where CUs are shared.
 *   - results are run-to-run deterministic (no floating-point atomics, fixed summation orders): every entry point is bit-reproducible.
 */
#ifndef GPTQ_MI355X_H
#define GPTQ_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPTQ_MI355X_ABI_VERSION 7
#define GPTQ_WORKSPACE_HEADER_BYTES 65536

typedef enum gptq_status_t {
    GPTQ_OK = 0,
    GPTQ_ERR_NULL = 1,         /* required pointer is NULL */
    GPTQ_ERR_SHAPE = 2,        /* K/N/M/group_size violate the layout rules */
    GPTQ_ERR_UNSUPPORTED = 3,  /* bits/dtype/mode combination not implemented by this entry point */
    GPTQ_ERR_WORKSPACE = 4,    /* workspace missing or smaller than gptq_workspace_bytes() */
    GPTQ_ERR_LAUNCH = 5        /* HIP reported a launch/runtime error */
} gptq_status_t;

typedef enum gptq_dtype_t { GPTQ_F16 = 0, GPTQ_BF16 = 1, GPTQ_F32 = 2 } gptq_dtype_t;

/* Zero-point convention:
 *   WRAP   z = (field + 1) & maxq   -- qlinear_cuda_old.py:301-304 and every native 4-bit kernel
 *   NOWRAP z =  field + 1           -- qlinear_cuda.py:262-264; also cuda_old's 3-bit branch */
typedef enum gptq_zero_mode_t { GPTQ_ZERO_WRAP = 0, GPTQ_ZERO_NOWRAP = 1 } gptq_zero_mode_t;

/* Fused epilogue of the caller.  SILU_MUL is the gate/up pair of a gated MLP stored as ONE layer
 * whose columns are [gate | up] (the reference concatenates packed tensors along out_features the same way,
 * fused_llama_attn.py:171-173):  out[M, N/2] = silu(y[:, :N/2]) * y[:, N/2:]  with y = x @ W (+ bias), silu and the
 * product evaluated on the fp32 sums and rounded once -- the arithmetic of the reference's fused MLP kernel
 * (fused_llama_mlp.py:237-239).  Needs N % 64 == 0. */
typedef enum gptq_epilogue_t { GPTQ_EPI_NONE = 0, GPTQ_EPI_SILU_MUL = 1 } gptq_epilogue_t;

/* One quantized linear layer.  POD; the caller owns every pointer and keeps it alive. */
typedef struct gptq_layer_t {
    const uint32_t *qweight;   /* [K/32*bits, N] */
    const uint32_t *qzeros;    /* [G, N/32*bits] */
    const void     *scales;    /* [G, N] dtype */
    const int32_t  *g_idx;     /* [K] or NULL (sequential) */
    const void     *bias;      /* [N] dtype or NULL */
    int32_t K, N, bits, group_size;   /* group_size > 0 (the module resolves -1 to K) */
    int32_t dtype;             /* gptq_dtype_t of x / scales / bias / out */
    int32_t zero_mode;         /* gptq_zero_mode_t */
    /* Optional derived (post_init) side buffers; all NULL = run straight from the checkpoint
     * tensors.  Built by gptq_make_sequential + gptq_resequence_qweight for act-order layers:
     * rows of qweight_seq are in group-sorted order and perm[i] is the original k of sorted
     * position i, so group(i) = i / group_size. */
    const uint32_t *qweight_seq;  /* [K/32*bits, N] or NULL */
    const int32_t  *perm;         /* [K] or NULL */
    int32_t epilogue;             /* gptq_epilogue_t; with SILU_MUL `out` is [M, N/2] */
    int32_t tiled_cols;           /* GPTQ_STRIP_COLS when the two side buffers below are given, else 0 */
    /* Optional derived (post_init) DECODE COPY of a 2-, 3-, 4- or 8-bit layer, built by gptq_prepack_decode (sizes: gptq_prepack_decode_bytes); both NULL =
     * decode streams the checkpoint layout.  With w[k][n] the layer's unpacked integers (rows taken from qweight_seq when the layer has one, else qweight;
     * k past K read as 0), KPL = 32 (16 at 8 bits) and WPL = 4 (3 at 3 bits, 2 at 2 bits):
     *   qweight_tiled [strip s of 16 columns][chunk c of 4 KPL k][k-slot kb 0..3][column 0..15][word 0..WPL-1]: the lane (kb, col) holds the KPL consecutive
     *                 k from k0 = 4 KPL c + KPL kb of column 16 s + col, re-encoded so that masking a word in place yields (k, k + 1) pairs in the order x lies:
     *                   4 bits  word w, stored nibble p = w[k0 + 8 w + {0,2,4,6,1,3,5,7}[p]]     (= a nibble shuffle of qweight[16 c + 4 kb + w][16 s + col])
     *                   8 bits  word w, stored byte p   = w[k0 + 4 w + {0,2,1,3}[p]]
     *                   3 bits  word j: pair p = 5 j + i (i = 0..4) at bit 3 i of the low (k0 + 2 p) and high (k0 + 2 p + 1) 16 bits; bit 15 / 31 = bit j of
     *                           w[k0 + 30] / w[k0 + 31] -- the checkpoint's word-straddling values are resolved here, once
     *                   2 bits  (round 6) word w: pair p (0..7) at bit 2 p of the low (k0 + 16 w + 2 p) and high (k0 + 16 w + 2 p + 1) 16 bits; read by the
     *                           decode kernel only (plain and act-order layers, up to 4 rows; no fused epilogue) -- the batched / prefill kernels keep the checkpoint rows
     *                 A strip is one contiguous run, a chunk one contiguous 1024 (3 bits: 768, 2 bits: 512) bytes = one wave load;
     *   qconst_tiled  [strip][group g][REC bytes] = 16 scales (layer dtype) at byte 0, then from byte 32 the 16 zero-points AS USED (zero_mode applied):
     *                 uint8 each, REC = 48, at 2 / 3 / 4 bits; uint16 each, REC = 64, at 8 bits (no-wrap reaches 256).
     * The reference re-lays its weights at load time in every fast backend (exllamav2 q_matrix.cu:19-42,149; exllama q4_matrix.cu:105-169;
     * marlin_repack.cu:8-92 + the scale permutation of qlinear_marlin.py:133-176), in place; here the checkpoint tensors are left as they are. */
    const uint32_t *qweight_tiled;
    const void     *qconst_tiled;
} gptq_layer_t;

#define GPTQ_STRIP_COLS 16

/* Kernel families a caller may force with gptq_tuning_t.path ("does not fit" is then an error instead of a silent fallback). */
typedef enum gptq_path_t {
    GPTQ_PATH_AUTO = 0,           /* the planner's choice */
    GPTQ_PATH_GEMV_GENERIC = 1,   /* fp32-math GEMV: any bits / dtype / raw act-order g_idx */
    GPTQ_PATH_GEMV_LDS = 2,       /* RETIRED in round 6 (round-1 LDS-staged comparison GEMV): GPTQ_ERR_UNSUPPORTED */
    GPTQ_PATH_GEMM = 3,           /* the MFMA GEMMs (tiled, wide, strips, 17..256-row kernel: the planner picks among them) */
    GPTQ_PATH_GEMV_DIRECT = 4,    /* RETIRED in round 6 (v_dot2 comparison GEMV): GPTQ_ERR_UNSUPPORTED */
    GPTQ_PATH_GEMV_MFMA = 5,      /* matrix-core GEMV on the checkpoint layout (4-bit fp16 / bf16 kernel, or the 2/3/8-bit one) */
    GPTQ_PATH_GEMV_STREAM = 6,    /* streamed (LDS-DMA) GEMV on the checkpoint layout */
    GPTQ_PATH_GEMV_DECODE_COPY = 8 /* decode kernel on the load-time decode copy (qweight_tiled / qconst_tiled): M <= 8; the planner's own choice up to 4 rows, and at
                                    * 5..8 rows on single plain 4-bit layers with K and N in 2048..4096 */
} gptq_path_t;

/* Optional launch-shape override; NULL = the planner's choice, which is what a drop-in caller passes.  lanes_n / waves / ksplit / path force a documented
 * geometry or kernel family.  reserved[] must be zero for a drop-in caller: the measurement tools under tools/ and the forced-geometry grids of the test
 * suite use these four words as LAB switches (A/B runs, ablations); their slots and values are named in include/gptq_mi355x_lab.h and are not part of
 * the drop-in contract. */
typedef struct gptq_tuning_t {
    int32_t lanes_n;     /* checkpoint-layout GEMVs: lanes of a wave laid along N (4, 8, 16, 64); 4 columns per lane */
    int32_t waves;       /* waves per workgroup (1..16) */
    int32_t ksplit;      /* workgroups along K (1 = no cross-workgroup reduction) */
    int32_t path;        /* gptq_path_t */
    int32_t reserved[4]; /* 0; lab switches: gptq_mi355x_lab.h */
} gptq_tuning_t;

int         gptq_abi_version(void);
const char *gptq_last_error(void);
const char *gptq_status_string(int status);

/* Once per device (the calling thread's current HIP device), outside stream capture, before the first forward on it:
 * grants the dynamic-LDS size of the kernels that use more than the 64 KiB default.  Idempotent.  The reference's
 * equivalents are the one-time per-device set-up calls of its backends (exllama_ext.cpp:100-131 prepare_buffers /
 * set_tuning_params; exllamav2 ext.cpp:26-93 make_q_matrix's temp_dq hand-over). */
int gptq_init(void);

/* Bytes of scratch gptq_forward/gptq_gemv/gptq_gemm may need for this layer and M (0 possible). */
size_t gptq_workspace_bytes(const gptq_layer_t *layer, int M);
/* Same for an explicit launch shape (what gptq_forward_ex/gptq_gemv/gptq_gemm need with `tuning`). */
size_t gptq_workspace_bytes_ex(const gptq_layer_t *layer, int M, const gptq_tuning_t *tuning);

/* max over M = 1..max_M of gptq_workspace_bytes(layer, M): the need is not monotone in M (K splits appear and disappear as
 * the planner changes kernels), so a caller that sizes ONE scratch buffer before hipGraph capture asks for this. */
size_t gptq_workspace_bytes_max(const gptq_layer_t *layer, int max_M);

/* out[M,N] = x[M,K] @ dequant(layer) (+ bias).  Picks GEMV (small M) or MFMA GEMM. */
int gptq_forward(const gptq_layer_t *layer, const void *x, void *out, int M,
                 void *workspace, size_t workspace_bytes, void *stream);

/* n_layers independent layers that read the SAME x[M, K] -- q/k/v of an attention block, gate/up of a gated MLP -- in one
 * call: outs[i] is [M, layers[i]->N].  The reference fuses such layers by concatenating their packed tensors along
 * out_features into one QuantLinear (fused_llama_attn.py:171-203, fused_llama_mlp.py:157-242); this entry point gives the
 * same single launch (<= 4 layers: up to 4 rows the decode-copy kernel over the column strips of all layers -- 3- / 4- / 8-bit fp16 / bf16 layers that
 * carry qweight_tiled, all plain or all act-order; 5 to 128 rows the batched-decode / 17..128-row kernels on plain 4-bit layers where the planner
 * prefers them) without touching or copying the checkpoint tensors, and runs the layers one after the other in every other case -- except that
 * act-order layers which share ONE `perm` pointer (q / k / v, gate / up of a GPTQ checkpoint: the order comes from their common input) read ONE
 * permuted x per call.  CONTRACT of a shared `perm` pointer: the layers that carry it have IDENTICAL g_idx, and every one's qweight_seq was built with
 * that permutation (gptq_make_sequential + gptq_resequence_qweight) -- pointer equality is all this entry point checks; a caller that reuses one perm
 * buffer for layers with different activation orders gets wrong results (the Python side compares the g_idx tensors once: share_act_order).
 * Results are those of n gptq_forward calls either way (same values within fp rounding: the K split may differ).
 * Workspace: gptq_workspace_bytes_multi. */
size_t gptq_workspace_bytes_multi(const gptq_layer_t *const *layers, int n_layers, int M);
int gptq_forward_multi(const gptq_layer_t *const *layers, int n_layers, const void *x, void *const *outs, int M,
                       void *workspace, size_t workspace_bytes, void *stream);
/* ... with an explicit launch shape for the one-launch kernel (experiments; tuning.path = 6 -- streamed GEMV -- or
 * tuning.path = 3 with tuning.reserved[2] = 4 -- batched-decode kernel -- or 5 -- the 17..128-row kernel -- make "does not fit" an error). */
size_t gptq_workspace_bytes_multi_ex(const gptq_layer_t *const *layers, int n_layers, int M, const gptq_tuning_t *tuning);
int gptq_forward_multi_ex(const gptq_layer_t *const *layers, int n_layers, const void *x, void *const *outs, int M,
                          void *workspace, size_t workspace_bytes, void *stream, const gptq_tuning_t *tuning);

/* The gated MLP of a decoder block, out[M, N] = down( silu(gate(x)) * up(x) ), as ONE call: replaces the reference's fused MLP caller
 * (auto_gptq/nn_modules/fused_llama_mlp.py:157-242: FusedLlamaMLPForQuantizedModel.forward = one fused gate|up kernel with the SiLU * mul
 * inside + c_proj).  gate and up are [K -> I], down is [I -> N]; plain layers (epilogue NONE), checkpoint tensors untouched; any bits / dtype /
 * act-order the single-layer entry points take.  Default: gate and up through gptq_forward_multi (one launch for decode rows) into two staging
 * buffers in the workspace, SiLU * mul (silu and the product on fp32, rounded once to the layer dtype, as fused_llama_mlp.py:237-239), down.
 * gptq_mlp_forward_ex takes no path override (round 3's one-launch persistent kernel was measured slower than these three steps and is a lab now:
 * tools/lab/mlp_ring.hip); it exists for signature symmetry with the other _ex entry points. */
size_t gptq_workspace_bytes_mlp(const gptq_layer_t *gate, const gptq_layer_t *up, const gptq_layer_t *down, int M);
size_t gptq_workspace_bytes_mlp_ex(const gptq_layer_t *gate, const gptq_layer_t *up, const gptq_layer_t *down, int M, const gptq_tuning_t *tuning);
int gptq_mlp_forward(const gptq_layer_t *gate, const gptq_layer_t *up, const gptq_layer_t *down, const void *x, void *out, int M,
                     void *workspace, size_t workspace_bytes, void *stream);
int gptq_mlp_forward_ex(const gptq_layer_t *gate, const gptq_layer_t *up, const gptq_layer_t *down, const void *x, void *out, int M,
                        void *workspace, size_t workspace_bytes, void *stream, const gptq_tuning_t *tuning);
/* Host-only: "kernel=unfused launches=3+ steps=..." for (gate, up, down, M, tuning); as gptq_describe_plan. */
int gptq_describe_mlp_plan(const gptq_layer_t *gate, const gptq_layer_t *up, const gptq_layer_t *down, int M, const gptq_tuning_t *tuning, char *out,
                           size_t out_bytes);

/* Same, with an explicit launch shape / path (tuning may be NULL). */
int gptq_forward_ex(const gptq_layer_t *layer, const void *x, void *out, int M,
                    void *workspace, size_t workspace_bytes, void *stream,
                    const gptq_tuning_t *tuning);

/* Memory-bound decode path (wavefront reductions); any M, intended for M <= 8. */
int gptq_gemv(const gptq_layer_t *layer, const void *x, void *out, int M,
              void *workspace, size_t workspace_bytes, void *stream, const gptq_tuning_t *tuning);

/* MFMA prefill path; fp16/bf16 only. */
int gptq_gemm(const gptq_layer_t *layer, const void *x, void *out, int M,
              void *workspace, size_t workspace_bytes, void *stream, const gptq_tuning_t *tuning);

/* W_out[K,N] (dtype) = scales[g(k),n] * (w[k,n] - z[g(k),n]); bit-exact vs the reference's
 * `weights` tensor (one rounding of the exact product). */
int gptq_dequant(const gptq_layer_t *layer, void *W_out, void *stream);

/* Integer unpack (bit-exact targets). w_out uint8 [K,N]; z_out int32 [G,N] (zero-point as used). */
int gptq_unpack_weights(const uint32_t *qweight, int K, int N, int bits, uint8_t *w_out, void *stream);
int gptq_unpack_zeros(const uint32_t *qzeros, int G, int N, int bits, int zero_mode, int32_t *z_out, void *stream);

/* Device pack(): intweight[k,n] = round((W[n,k] + zero[g,n]*scale[g,n]) / scale_cast[g,n]) packed
 * along K exactly like the reference (fields OR-ed unmasked).  W is [N,K] in w_dtype; scale_in /
 * zero_in are [G,N] in qparam_dtype (already transposed); scales_out [G,N] in w_dtype receives the
 * cast scales.  g_idx NULL = sequential. */
int gptq_pack_weights(const void *W, const void *scale_in, const void *zero_in, const int32_t *g_idx,
                      int K, int N, int bits, int group_size, int w_dtype, int qparam_dtype,
                      uint32_t *qweight_out, void *scales_out, void *stream);
int gptq_pack_zeros(const void *zero_in, int G, int N, int bits, int qparam_dtype,
                    uint32_t *qzeros_out, void *stream);

/* Act-order support.  HOST function: stable counting sort of k by g_idx (host pointers).
 * perm_out[i] = original k at sorted position i.  *uniform_out = 1 iff sorted position i has
 * group i / group_size for every i (then the fast kernels can use qweight_seq + perm). */
int gptq_make_sequential(const int32_t *g_idx_host, int K, int group_size,
                         int32_t *perm_out_host, int *uniform_out);
/* HOST function: every g_idx[k] must name an existing group, 0 <= g_idx[k] < G (G = rows of scales / qzeros).  The kernels
 * index scales / qzeros with raw g_idx values on act-order layers; the reference's torch indexing raises on a bad index
 * (qlinear_cuda.py:302 `self.scales[self.g_idx.long()]`), this is the same check for a C caller.  GPTQ_ERR_SHAPE on violation. */
int gptq_validate_g_idx(const int32_t *g_idx_host, int K, int G);
/* qweight_seq[row-order = perm] from qweight; device pointers. */
int gptq_resequence_qweight(const uint32_t *qweight, const int32_t *perm, int K, int N, int bits,
                            uint32_t *qweight_seq_out, void *stream);
/* The decode copy of a 2-, 3-, 4- or 8-bit layer (layouts: gptq_layer_t.qweight_tiled / qconst_tiled): exact integer re-arrangements of the packed fields, the
 * scales (bit copies) and the zero-points (the value the kernels subtract, zero_mode applied), written to caller-owned buffers of
 * gptq_prepack_decode_bytes() bytes.  Source: layer->qweight_seq when present, else layer->qweight; layer->qweight_tiled / qconst_tiled / tiled_cols
 * are ignored.  Needs bits 2 / 3 / 4 / 8 (2 bits: no fused epilogue), fp16 / bf16, group_size = KPL times a power of two (or group_size >= K), K % 32 == 0, N % 16 == 0, no raw act-order;
 * otherwise GPTQ_ERR_UNSUPPORTED (and sizes 0).  The role of the reference's load-time re-layouts -- exllamav2 shuffle_kernel
 * (exllamav2/cuda/q_matrix.cu:19-42, called :149), exllama make_sequential (exllama/cuda_func/q4_matrix.cu:105-169), Marlin's repack kernel
 * (marlin/marlin_repack.cu:8-92) -- without touching the checkpoint tensors. */
int gptq_prepack_decode_bytes(const gptq_layer_t *layer, size_t *tiled_bytes, size_t *const_bytes);
int gptq_prepack_decode(const gptq_layer_t *layer, uint32_t *qweight_tiled_out, void *qconst_tiled_out, void *stream);
/* The exact inverse of the weights half (round 5): qweight_out [K/32*bits, N] = the packed rows the copy was made from (qweight, or qweight_seq of an act-order
 * layer), bit for bit.  What lets ONE copy of the weights stay on the device -- the reference's fast backends re-lay their weights IN PLACE
 * (exllama/cuda_func/q4_matrix.cu:160, exllamav2/cuda/q_matrix.cu:149) -- while state_dict() / a re-save / the kernels that read rows are still served:
 * QuantLinear.post_init(release_checkpoint_layout=True) moves qweight to host memory and rebuilds rows into a shared scratch only for calls that need them. */
int gptq_unprepack_decode(const uint32_t *qweight_tiled, int K, int N, int bits, uint32_t *qweight_out, void *stream);
/* x_out[m, i] = x[m, perm[i]] */
int gptq_permute_columns(const void *x, const int32_t *perm, int M, int K, int dtype, void *x_out, void *stream);

/* Host-only introspection: which kernel and launch geometry gptq_forward_ex would use for (layer, M, tuning), written as a
 * short "key=value ..." line into out (NUL-terminated, truncated to out_bytes).  No device work, no pointer is dereferenced
 * except the struct fields; returns GPTQ_OK or the validation status gptq_forward_ex would return.  Examples:
 *   "path=gemv kernel=mfma ln=4 waves=16 u=2 ksplit=1 mt=1 strips=256 pair=0 perm=0"
 *   "path=gemm kernel=tiled mt=4 bk=64 kg=2 ksplit=1 tiles=16x16 perm=1 dma=1"
 * (the reference's dispatch thresholds, SURVEY §8 a15, are constants in its sources; here they are queryable.) */
int gptq_describe_plan(const gptq_layer_t *layer, int M, const gptq_tuning_t *tuning, char *out, size_t out_bytes);

/* ---- AWQ checkpoint ingest (4-bit only, as the reference: auto_gptq/modeling/_utils.py:525-701) ----------------
 * AWQ side: awq_qweight u32 [K, N/8] (nibble p of word c = column 8c + {0,2,4,6,1,3,5,7}[p]), awq_qzeros u32 [G, N/8]
 * (same order, raw zero-point, no -1), awq_scales fp16 [G, N] (= the GPTQ `scales` tensor, passed through unchanged).
 *
 * gptq_awq_unpack replaces unpack_awq (_utils.py:556-621): weight_kn_out fp16 [K, N] =
 *   half(w * s) - half(z * s)   (each product and the difference rounded to fp16 once, the reference's op order);
 *   the reference returns its transpose view [N, K].  zeros_out int8 [G, N] = the raw zero-points in natural column order.
 * gptq_awq_repack is unpack_awq + pack_from_tensors (_utils.py:624-701) as one integer pass: GPTQ qweight_out u32 [K/8, N]
 *   (nibble j of word r = w[8r + j, n]) and qzeros_out u32 [G, N/8] (field = (z - 1) & 15).  Identical to the reference's
 *   fp16 round trip whenever that round trip is faithful (finite, normal scales; tests/golden/awq_*.npz). */
int gptq_awq_unpack(const uint32_t *awq_qweight, const uint32_t *awq_qzeros, const void *awq_scales, int K, int N,
                    int group_size, void *weight_kn_out, int8_t *zeros_out, void *stream);
int gptq_awq_repack(const uint32_t *awq_qweight, const uint32_t *awq_qzeros, int K, int N, int group_size,
                    uint32_t *qweight_out, uint32_t *qzeros_out, void *stream);

/* ---- direct peer-store all-gather for out_features-parallel layers; experimental, off by default ----------
 * The reference runs a layer on one GPU only (tests/test_q4.py:1224-1226: `test_multigpu` is a TODO); the column split is the
 * property its fused q/k/v caller relies on (fused_llama_attn.py:171-186).  Rank r of T computes out[:, r*N/T : (r+1)*N/T]; the
 * exchange below replaces the ring all-gather behind it on a point-to-point fabric: every rank stores its [M, N/T] slice into the
 * exchange buffer of EVERY rank (one xGMI link per peer), raises a flag there, waits for the T flags of its own buffer and copies
 * the gathered [M, N] rows to `out`.
 *
 * The caller owns all memory: per rank two exchange buffers [rows_max, N] (they alternate by call parity), uint32
 * flags[GPTQ_PEER_MAX] and uint32 state[4], the last two zeroed once; the buffers and flags of the peers are mapped into the
 * caller's address space (hipIpcOpenMemHandle / the framework's IPC) and listed in gptq_peer_group_t in RANK order (entry `rank` =
 * the caller's own).  Across GPUs buffers and flags must be fine-grained allocations.  Calls are collective: every rank issues
 * the same sequence of gathers with the same M.  Two launches per gather, no host state per call (legal under hipGraph capture,
 * and a captured graph replays: the call epoch lives in state[0]).  state[3] != 0 afterwards = a wait gave up after
 * `max_spins` polls (a peer never arrived) and `out` is incomplete -- the kernels never spin unbounded. */
#define GPTQ_PEER_MAX 8
typedef struct gptq_peer_group_t {
    void     *xbuf[2][GPTQ_PEER_MAX];  /* xbuf[p][r]: exchange buffer of parity p of rank r, [rows_max, N] dtype */
    uint32_t *flags[GPTQ_PEER_MAX];    /* flags[r]: rank r's arrival flags, uint32[GPTQ_PEER_MAX] */
    uint32_t *state;                   /* this rank's uint32[4]: gathers completed, two tickets, timeout raised */
    int32_t world, rank;               /* 1 <= world <= GPTQ_PEER_MAX */
    int32_t rows_max, N;               /* geometry of the exchange buffers; N = out_features of the full layer */
} gptq_peer_group_t;

/* scatter: y_local [M, n_local] (n_local = N / world) -> columns [rank*n_local, (rank+1)*n_local) of every rank's exchange
 * buffer, then the arrival flags.  collect: wait for all ranks' slices of this call, copy [M, N] to `out`, advance the epoch.
 * gather = scatter + collect.  Independent work of the caller may be enqueued between scatter and collect -- ON THE SAME STREAM: every collect
 * (re-)publishes this rank's arrival flag of its epoch when it starts (that is what lets gptq_forward_scatter go without a flag of its own), which is only
 * correct behind the rank's payload stores in stream order.  A scatter on one stream and its collect on another -- or a collect enqueued first -- raises
 * the flag over an incomplete payload and the peers copy stale rows without any error. */
int gptq_peer_scatter(const gptq_peer_group_t *pg, const void *y_local, int M, int n_local, int dtype, void *stream);
int gptq_peer_collect(const gptq_peer_group_t *pg, void *out, int M, int dtype, uint32_t max_spins, void *stream);
int gptq_peer_gather(const gptq_peer_group_t *pg, const void *y_local, void *out, int M, int n_local, int dtype,
                     uint32_t max_spins, void *stream);
/* The scatter as the EPILOGUE of the rank's decode kernel (round 4): layer = this rank's column shard (layer->N = pg->N / world) carrying its decode copy,
 * M <= 4.  The strip owners of the decode-copy kernel store their [M][16] outputs straight into every rank's exchange buffer (16-byte system-scope
 * write-through stores) -- no y_local, no scatter launch: a tensor-parallel layer is the local kernel + ONE collect.  The rank's arrival flag is raised by
 * the first block of its gptq_peer_collect launch (every collect (re-)publishes the flag of its epoch before it waits: behind this kernel in stream order
 * the payload is complete); gptq_peer_publish raises it on its own -- needed only when several ranks share ONE stream (simulations: each rank's collect
 * would wait for flags that later launches of the same stream raise) or when unrelated work sits between the scatter and the collect.
 * workspace as for gptq_forward (gptq_workspace_bytes(layer, M)).  GPTQ_ERR_UNSUPPORTED when the layer / M does not qualify (then: gptq_forward +
 * gptq_peer_scatter).  gptq_forward_gather = gptq_forward_scatter + gptq_peer_collect.  Same collective contract as gptq_peer_gather. */
int gptq_forward_scatter(const gptq_layer_t *layer, const void *x, int M, const gptq_peer_group_t *pg, void *workspace, size_t workspace_bytes,
                         void *stream);
int gptq_forward_gather(const gptq_layer_t *layer, const void *x, void *out, int M, const gptq_peer_group_t *pg, uint32_t max_spins, void *workspace,
                        size_t workspace_bytes, void *stream);
int gptq_peer_publish(const gptq_peer_group_t *pg, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GPTQ_MI355X_H */
