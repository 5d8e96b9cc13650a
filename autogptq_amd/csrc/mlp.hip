// mlp.hip -- what the gated-MLP entry point (gptq_mlp_forward, capi.hip) needs besides the linear kernels: the SiLU * mul between [gate | up] and down,
// and the per-device CU count gptq_init() records (the kernels that wait inside a launch for sibling workgroups check their grid against it).
// Replaces (reference): the elementwise part of FusedLlamaMLPForQuantizedModel.forward, auto_gptq/nn_modules/fused_llama_mlp.py:237-239.
// (Round 3's one-launch persistent MLP kernel lived here; measured slower than the three launches it replaced, it is a lab now: tools/lab/mlp_ring.hip,
//  DESIGN.md section 4.1c.)
#include "common.cuh"
#include "launch.h"

namespace gptq {
namespace mlpk {

template <typename T>
__global__ void __launch_bounds__(256) silu_mul2_kernel(const T* __restrict__ g, const T* __restrict__ u, T* __restrict__ out, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float a = DType<T>::to_f32(g[i]), b = DType<T>::to_f32(u[i]);
        out[i] = DType<T>::from_f32(a / (1.f + __expf(-a)) * b);
    }
}

int g_cu_count[64] = {0};          // per device ordinal, filled by init_mlp_device (gptq_init); 0 = not initialised

}  // namespace mlpk
using namespace mlpk;

hipError_t launch_silu_mul2(const void* g, const void* u, void* out, size_t total, int dtype, hipStream_t st) {
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    switch (dtype) {
        case GPTQ_F16: hipLaunchKernelGGL(silu_mul2_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const f16*)g, (const f16*)u, (f16*)out, total); break;
        case GPTQ_BF16: hipLaunchKernelGGL(silu_mul2_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)g, (const bf16*)u, (bf16*)out, total); break;
        default: hipLaunchKernelGGL(silu_mul2_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)g, (const float*)u, (float*)out, total);
    }
    return hipGetLastError();
}

hipError_t init_mlp_device() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) g_cu_count[dev] = cus;
    return e;
}

}  // namespace gptq
