// gemm_rows_b38.hip -- the 3-bit and 8-bit instantiations of the exchange-free batched-decode kernel (gemm_rows.hip / gemm_rows_kernel.cuh): BASELINE config 5's
// batched rows (int3 / int8, group_size 32), which ran on the rounds 2 - 3 kernels.  A translation unit of its own for the build time.
#include "gemm_rows_kernel.cuh"

namespace gptq {

hipError_t init_gemm_rows_b3_device();                        // gemm_rows_b3.hip
hipError_t launch_gemm_rows_b3(int dtype, int gm, const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st);

hipError_t init_gemm_rows_b38_device() {
    hipError_t e = init_gemm_rows_b3_device();
    if (e == hipSuccess) e = rows_grant_bits<8>();
    return e;
}

hipError_t launch_gemm_rows_b38(int bits, int dtype, int gm, const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) {
    if (bits == 3) return launch_gemm_rows_b3(dtype, gm, pl, p, st);
    if (bits == 8) return rows_launch_bits<8>(dtype, gm, pl, p, st);
    return hipErrorInvalidValue;
}

}  // namespace gptq
