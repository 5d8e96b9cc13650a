// gemv_tiled_pair.hip -- the decode-copy kernel for [gate | up] layers with the fused SiLU * mul epilogue (gemv_tiled_kernel.cuh, XM = 4: a workgroup streams
// strip s of the gate half and strip s of the up half behind one staged x).  Replaces for decode rows: the reference's fused MLP
// (auto_gptq/nn_modules/fused_llama_mlp.py:131-306).  A translation unit of its own for build time -- and so that the plain kernels carry none of it.
#include "gemv_tiled_kernel.cuh"

namespace gptq {

hipError_t launch_tiled_pair(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st) {
    return dtype == GPTQ_BF16 ? launch_tiled_bits<bf16, 4>(pl, p, st) : launch_tiled_bits<f16, 4>(pl, p, st);
}
hipError_t init_gemv_tiled_pair_device() { return grant_tiled_lds<4>(); }

}  // namespace gptq
