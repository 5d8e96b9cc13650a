// gemm.hip -- MFMA prefill path (placeholder until the kernel lands)
#include "common.cuh"
#include "launch.h"
namespace gptq {
GemmPlan plan_gemm(const gptq_layer_t&, int, const gptq_tuning_t*) { return GemmPlan{false, false, 0, 0, 0}; }
hipError_t launch_gemm(const gptq_layer_t&, const GemmPlan&, const void*, void*, int, void*, hipStream_t) { return hipErrorNotSupported; }
}
