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

// SiLU * mul AND the x permute of an act-order `down` in ONE pass (round 6): out[m][i] = silu(g[m][perm[i]]) * u[m][perm[i]].  gptq_mlp_forward used to run the
// elementwise pass and then down's own permute pre-pass (permute_rows4_kernel, gemm.hip: 23 us of the 171 of an 11008 -> 4096 act-order layer call at 2048 rows,
// and a second round trip of the [M, I] activations through memory); the reference's fused MLP hands its c_proj the activations in the order that layer's rows
// are stored in (fused_llama_mlp.py:131-306; exllama fuses the column map into its consumer, exllama/cuda_func/q4_matmul.cu:92-126).  Same structure as
// permute_rows4_kernel<false>: FOUR rows per workgroup held interleaved in LDS ([k][4 rows], 8 bytes per k) so that one ds_read_b64 per index fetches all four
// rows' values; the load phase computes silu(g) * u on fp32 and rounds once (the arithmetic of silu_mul2_kernel: the same bits as the two passes).
template <typename T>
__global__ void __launch_bounds__(1024) silu_mul2_permute_rows4_kernel(const unsigned short* __restrict__ g, const unsigned short* __restrict__ u, const int* __restrict__ perm,
                                                                      int M, int K, unsigned short* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];                // [K][4] values
    const int m0 = blockIdx.x * 4;
    size_t ro[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ro[r] = (size_t)min(m0 + r, M - 1) * K;
    for (int i = threadIdx.x * 8; i < K; i += (int)blockDim.x * 8) {
        u32x4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u32x4 gv = *(const u32x4*)(g + ro[r] + i), uv = *(const u32x4*)(u + ro[r] + i);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned pk = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a = DType<T>::to_f32(__builtin_bit_cast(T, (unsigned short)(gv[w] >> (16 * h))));
                    const float b = DType<T>::to_f32(__builtin_bit_cast(T, (unsigned short)(uv[w] >> (16 * h))));
                    const T y = DType<T>::from_f32(a / (1.f + __expf(-a)) * b);
                    pk |= (unsigned)__builtin_bit_cast(unsigned short, y) << (16 * h);
                }
                v[r][w] = pk;
            }
        }
        u32x4 o[4];                                                             // 8 k x {rows 0 | 1, rows 2 | 3}
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            o[w][0] = __builtin_amdgcn_perm(v[1][w], v[0][w], 0x05040100u);     // k = 2 w:     row 0 | row 1 << 16
            o[w][1] = __builtin_amdgcn_perm(v[3][w], v[2][w], 0x05040100u);     //              row 2 | row 3 << 16
            o[w][2] = __builtin_amdgcn_perm(v[1][w], v[0][w], 0x07060302u);     // k = 2 w + 1
            o[w][3] = __builtin_amdgcn_perm(v[3][w], v[2][w], 0x07060302u);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) *(u32x4*)(smem + (size_t)i * 8 + w * 16) = o[w];
    }
    __syncthreads();
    for (int i = threadIdx.x * 8; i < K; i += (int)blockDim.x * 8) {
        const u32x4 p0 = *(const u32x4*)(perm + i), p1 = *(const u32x4*)(perm + i + 4);
        u32x2 gg[8];                                                            // gg[j] = the four rows' values at source index j of this piece
#pragma unroll
        for (int j = 0; j < 4; ++j) { gg[j] = *(const u32x2*)(smem + (size_t)p0[j] * 8); gg[4 + j] = *(const u32x2*)(smem + (size_t)p1[j] * 8); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m0 + r >= M) break;
            const unsigned sel = (r & 1) ? 0x07060302u : 0x05040100u;
            const int h = r >> 1;
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) o[w] = __builtin_amdgcn_perm(gg[2 * w + 1][h], gg[2 * w][h], sel);
            *(u32x4*)(out + (size_t)(m0 + r) * K + i) = o;                      // plain stores: the GEMM behind this pass reads them at once (gemm.hip)
        }
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

// silu(g) * u gathered through perm into out (fp16 / bf16; K % 8 == 0 and 8 K bytes of LDS): hipErrorInvalidValue where the form does not apply (the caller then
// runs the two passes)
bool silu_mul2_permute_ok(int K, int dtype) { return (dtype == GPTQ_F16 || dtype == GPTQ_BF16) && K % 8 == 0 && (size_t)K * 8 <= 160 * 1024; }
hipError_t launch_silu_mul2_permute(const void* g, const void* u, const int32_t* perm, int M, int K, int dtype, void* out, hipStream_t st) {
    if (!silu_mul2_permute_ok(K, dtype) || M <= 0) return hipErrorInvalidValue;
    const dim3 grid((M + 3) / 4), block(K >= 8192 ? 1024 : 512);      // one workgroup per CU at K = 11008 (88 KiB of LDS): 16 waves keep enough loads in flight
    if (dtype == GPTQ_F16)
        hipLaunchKernelGGL(silu_mul2_permute_rows4_kernel<f16>, grid, block, (size_t)K * 8, st, (const unsigned short*)g, (const unsigned short*)u, perm, M, K, (unsigned short*)out);
    else
        hipLaunchKernelGGL(silu_mul2_permute_rows4_kernel<bf16>, grid, block, (size_t)K * 8, st, (const unsigned short*)g, (const unsigned short*)u, perm, M, K, (unsigned short*)out);
    return hipGetLastError();
}

hipError_t init_mlp_device() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)silu_mul2_permute_rows4_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)silu_mul2_permute_rows4_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) g_cu_count[dev] = cus;
    return e;
}

}  // namespace gptq
