// gemv_tiled_act.hip -- the act-order instantiations of the decode-copy kernel (gemv_tiled_kernel.cuh): x gathered through the layer's perm while it is
// staged, weights from the decode copy of the re-sequenced rows.  A translation unit of its own only for build time.
#include "gemv_tiled_kernel.cuh"

namespace gptq {

hipError_t launch_tiled_act(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st) {
    return dtype == GPTQ_BF16 ? launch_tiled_bits<bf16, 1>(pl, p, st) : launch_tiled_bits<f16, 1>(pl, p, st);
}
hipError_t init_gemv_tiled_act_device() { return grant_tiled_lds<1>(); }

}  // namespace gptq
