// gemm_rows_b3.hip -- the 3-bit instantiations of the exchange-free batched-decode kernel (gemm_rows.hip / gemm_rows_kernel.cuh); a translation unit of its own
// for the build time (gemm_rows_b38.hip holds the 8-bit ones and the dispatch over the two widths).
#include "gemm_rows_kernel.cuh"

namespace gptq {

hipError_t init_gemm_rows_b3_device() { return rows_grant_bits<3>(); }
hipError_t launch_gemm_rows_b3(int dtype, int gm, const RowsPlan& pl, const rowsk::RowsParams& p, hipStream_t st) { return rows_launch_bits<3>(dtype, gm, pl, p, st); }

}  // namespace gptq
