// lab_launch.h -- declarations of the lab kernels kept under tools/lab/ (measured negative results: not linked into libgptq_mi355x.so).
#pragma once
#include "launch.h"

namespace gptq {

// Fused gated MLP (mlp.hip): gate | up -> SiLU * mul -> down in ONE persistent launch, one row of x.
struct MlpPlan {
    bool ok;                 // the three layers qualify (plain 4-bit fp16/bf16, shared group size, M = 1) and the geometry fits one workgroup per CU
    int ns, nwg, gshift;     // ring slots per wave, workgroups (= CUs of the device), log2(packed rows per group)
    int qA, rA, qB, rB;      // column chunks per workgroup: q (+ 1 for the first r workgroups), gate/up and down
    int lrpiA, lrpiB;        // log2(packed rows per DMA instruction) in the gate/up and the down panels
    int lspA, lszA, lspB, lszB;   // log2 of the padded LDS table rows (scales, zero-point words)
    int off_x, off_cst, off_red, off_ctr;
    size_t lds_bytes;
    size_t exchange_bytes;   // behind the header: the activation granules [I / 2] x 8 bytes
};
MlpPlan plan_mlp(const gptq_layer_t& gate, const gptq_layer_t& up, const gptq_layer_t& down, int M, int nwg_override);
hipError_t launch_mlp(const gptq_layer_t& gate, const gptq_layer_t& up, const gptq_layer_t& down, const MlpPlan& pl, const void* x, void* out,
                      void* ws_header, void* exchange, hipStream_t st);
// gemm_ldsb.hip: prefill kernel with the dequantised weights shared through LDS (4-bit fp16/bf16, 256 x 128 tiles)
bool ldsb_supported(const gptq_layer_t& L, int M);
hipError_t init_gemm_ldsb_device();
hipError_t launch_gemm_ldsb(const gptq_layer_t& L, const uint32_t* qweight, const void* x, void* out, int M, hipStream_t st, int bk = 0, int kgroups = 0,
                            int abl = 0);

}  // namespace gptq
