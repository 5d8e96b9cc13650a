// gemv_tiled_multi.hip -- the decode-copy kernel with FOUR / TWO adjacent strips per workgroup behind one staged x (gemv_tiled_kernel.cuh, XM = 5 / 6): the
// 3..8-row forms on layers of many strips, where every 16-column strip staging its own rows of x was what the time grew with.  A translation unit of its own
// for build time -- and so that the plain kernels carry none of it.
#include "gemv_tiled_kernel.cuh"

namespace gptq {

hipError_t launch_tiled_multi(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st) {
    if (pl.nstr == 4) return dtype == GPTQ_BF16 ? launch_tiled_bits<bf16, 5>(pl, p, st) : launch_tiled_bits<f16, 5>(pl, p, st);
    if (pl.nstr == 2) return dtype == GPTQ_BF16 ? launch_tiled_bits<bf16, 6>(pl, p, st) : launch_tiled_bits<f16, 6>(pl, p, st);
    return hipErrorInvalidValue;
}
hipError_t init_gemv_tiled_multi_device() {
    hipError_t e = grant_tiled_lds<5>();
    hipError_t e2 = grant_tiled_lds<6>();
    return e != hipSuccess ? e : e2;
}

}  // namespace gptq
