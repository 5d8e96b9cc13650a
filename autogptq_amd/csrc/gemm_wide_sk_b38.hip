// gemm_wide_sk_b38.hip -- the stream-K prefill kernel (gemm_wide_sk.hip) on the 3-bit and 8-bit decode copies and on 32-wide groups: BASELINE config 5's prefill rows,
// which ran on the 128 x 256 BK = 32 row kernel (gemm.hip) at 0.31 - 0.33 of the dense bf16 peak where the 4-bit g128 layer reaches 0.42.  Same schedule, same
// x path, same exchange; the weights come from the copy's 3-word lanes (utils.hip: prepack_decode_weights_kernel<3>) and the constants from its records
// (qconst_tiled: the zero-point as used, so no wrap mask in the loop).  Reference behaviour: qlinear_cuda_old.py:203-262 dequantises every width to fp16 and
// calls one matmul, so the prefill throughput of a 3-bit layer equals the 4-bit one's there; this file is what makes that true here.
#include <type_traits>

#include "common.cuh"
#include "gemm_wide_common.cuh"
#include "gemm_wide_sk_kernel.cuh"
#include "launch.h"

namespace gptq {
namespace wide {

template <typename T, int BITS, int GM>
__global__ void __launch_bounds__(256, 1) gemm_wide_skb_kernel(WskParams p) {
    wsk_body<T, BITS, GM>(p);
}

}  // namespace wide

template <typename T, int BITS, int GM>
static hipError_t grant_one() {
    return hipFuncSetAttribute((const void*)wide::gemm_wide_skb_kernel<T, BITS, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, wide::WSK_LDS_BYTES);
}
template <typename T>
static hipError_t grant_all() {
    hipError_t e = grant_one<T, 4, 2>();
    if (e == hipSuccess) e = grant_one<T, 3, 0>();
    if (e == hipSuccess) e = grant_one<T, 3, 1>();
    if (e == hipSuccess) e = grant_one<T, 3, 2>();
    if (e == hipSuccess) e = grant_one<T, 8, 0>();
    if (e == hipSuccess) e = grant_one<T, 8, 1>();
    if (e == hipSuccess) e = grant_one<T, 8, 2>();
    return e;
}
hipError_t init_gemm_wide_skb_device() {
    hipError_t e = grant_all<f16>();
    if (e == hipSuccess) e = grant_all<bf16>();
    return e;
}

template <typename T, int BITS, int GM>
static void launch_one(dim3 grid, dim3 block, hipStream_t st, const wide::WskParams& p) {
    hipLaunchKernelGGL((wide::gemm_wide_skb_kernel<T, BITS, GM>), grid, block, wide::WSK_LDS_BYTES, st, p);
}
template <typename T>
static void launch_t(int bits, int gm, dim3 grid, dim3 block, hipStream_t st, const wide::WskParams& p) {
    if (bits == 4) launch_one<T, 4, 2>(grid, block, st, p);
    else if (bits == 8) {
        if (gm == 0) launch_one<T, 8, 0>(grid, block, st, p);
        else if (gm == 1) launch_one<T, 8, 1>(grid, block, st, p);
        else launch_one<T, 8, 2>(grid, block, st, p);
    } else if (gm == 0) launch_one<T, 3, 0>(grid, block, st, p);
    else if (gm == 1) launch_one<T, 3, 1>(grid, block, st, p);
    else launch_one<T, 3, 2>(grid, block, st, p);
}
void launch_gemm_wide_skb(int bits, int gm, int dtype, dim3 grid, dim3 block, hipStream_t st, const wide::WskParams& p) {
    if (dtype == GPTQ_F16) launch_t<f16>(bits, gm, grid, block, st, p);
    else launch_t<bf16>(bits, gm, grid, block, st, p);
}

}  // namespace gptq
