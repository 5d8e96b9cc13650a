// gemv_tiled_peer.hip -- the decode-copy kernel with the tensor-parallel epilogue (gemv_tiled_kernel.cuh, XM = 3: gptq_forward_scatter): the strip owners
// store their outputs into every rank's exchange buffer.  A translation unit of its own for build time -- and so that the plain kernels carry none of it.
#include "gemv_tiled_kernel.cuh"

namespace gptq {

hipError_t launch_tiled_peer(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st) {
    return dtype == GPTQ_BF16 ? launch_tiled_bits<bf16, 3>(pl, p, st) : launch_tiled_bits<f16, 3>(pl, p, st);
}
hipError_t init_gemv_tiled_peer_device() { return grant_tiled_lds<3>(); }

}  // namespace gptq
