// mlp.hip -- the gated MLP of a decoder block in ONE launch for one row of x (decode):
//     out = down( silu(gate(x)) * up(x) )          three plain 4-bit QuantLinears, checkpoint tensors read in place
// Reference role: auto_gptq/nn_modules/fused_llama_mlp.py:157-242 (FusedLlamaMLPForQuantizedModel: one fused gate|up kernel with the
// SiLU * mul inside, then c_proj as a second kernel); the decode GEMVs behind it are autogptq_cuda_kernel_256.cu:1367-1437 /
// exllamav2 q_gemm_kernel_gptq.cuh:39-194.
//
// Why one launch: a finished decode launch costs ~2.9 us + bytes / 4.4 TB/s -- the dependent kernel boundary, the ramp of
// an empty memory system and the drain are paid per launch, and the down projection can not start its weight stream before gate|up
// has finished although its WEIGHTS depend on nothing.  Here:
//   * the grid is PERSISTENT and BALANCED: one 16-wave workgroup per CU (grid = CU count), and workgroup w owns the w-th 1/grid of
//     the 16-byte column chunks of gate AND up (the same chunk range of both: silu(g) * u is local) and later the w-th 1/grid of the
//     column chunks of down -- 10.75 chunks of 2752 on 256 CUs is 11 or 10, a 2 % imbalance where whole 64-column strips give 2.7 rounds;
//   * weights stream global -> LDS by DMA (global_load_lds_dwordx4 nt, 1 KiB per wave instruction) through a per-wave RING of NS
//     slots that is refilled as soon as a slot has been read: every wave keeps NS KiB in flight for the whole launch, and the ring
//     runs straight on from the gate/up panels into the wave's rows of `down` -- the down weights are in flight / landed while the
//     activation is still being exchanged;
//   * x, the scales and the zero-points reach LDS by DMA too (issued BEFORE the weight DMAs: they return first), so the only VMEM
//     operations in the loop are the ring's own DMAs and s_waitcnt vmcnt(NS - 1) is exact;
//   * every lane is independent: lane l of a DMA instruction holds (row = l / cw, chunk = l % cw) of the panel, 8 k x 4 columns; the
//     k reduction runs on v_mfma_f32_4x4x4 with the lane's own x slice as its A row -- register (l & 3) of the lane's accumulators
//     (the block diagonal) is its dot product, the other three are cross terms nobody reads;
//   * the activation crosses the chip once, as 8-byte {2 x T, tag} granules written with write-through stores and validated by the
//     readers themselves (tag = launch epoch in a NaN pattern): no flag, no fence, no ordering assumption; every spin is bounded.
// The arithmetic is the decode kernels' (exact w - z in packed fp16, fp32 group sums, scale on the fp32 sums, silu and the product
// on fp32, one rounding to T for the activation and one for the output).
#include <algorithm>
#include <type_traits>
#include <utility>

#include "common.cuh"
#include "launch.h"
#include "lab_launch.h"

namespace gptq {

namespace mlpk {      // named (not anonymous) so that rocprof traces show gptq::mlpk::mlp_ring_kernel instead of gptq::_GLOBAL__N_1

constexpr int MLP_W = 16;          // waves per workgroup (1024 threads: one workgroup per CU)
constexpr int MLP_CWMAX = 16;      // column chunks (of 4 columns) per workgroup and panel: a DMA instruction covers 64 / cw rows

struct MlpLayerArgs {
    const unsigned* qweight;
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
};
struct MlpParams {
    MlpLayerArgs gate, up, down;
    const void* x;                 // [K]
    void* out;                     // [N]
    unsigned* hdr;                 // header tail: [0] launch epoch, [1] workgroups done, [2] sticky error (a bounded wait gave up), [4:5] lab timeline buffer
    unsigned long long* gran;      // [I / 2] exchange granules {lo: act[2m] | act[2m+1] << 16, hi: tag}
    int K, I, N;                   // hidden -> intermediate -> hidden
    int gshift;                    // log2(packed rows per group)
    int zero_mode;
    int nwg;
    unsigned max_spins;
    // work split (host-computed: no division in the kernel): workgroup w owns chunks [w * q + min(w, r), + q + (w < r)) of a panel
    int qA, rA, qB, rB;
    int lrpiA, lrpiB;              // log2(packed rows per DMA instruction) of the gate/up and the down panels (launch-uniform, <= gshift)
    int lspA, lszA, lspB, lszB;    // log2 of the padded row lengths (dwords) of the scale / zero-point tables in LDS
    // LDS byte offsets (host-computed): ring at 0
    int off_x, off_cst, off_red, off_ctr;
};

// ---- small device helpers ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16_nt_(const void* gsrc, unsigned lds_dst) {            // 16 B per lane, nontemporal
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16_(const void* gsrc, unsigned lds_dst) {               // 16 B per lane, default policy
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4_(const void* gsrc, unsigned lds_dst) {                // 4 B per lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// vreg[lane_sel] = val (both wave-uniform).  No builtin in this clang; the s_nop covers the VALU-wrote-SGPR -> lane-select hazard, which the
// compiler's hazard recognizer cannot see inside asm.
__device__ __forceinline__ int writelane(int val, int lane_sel, int vreg) {
    unsigned keep;                                                      // lane select through M0: one SGPR operand per VALU instruction on gfx9
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(vreg), "=&s"(keep) : "s"(val), "s"(lane_sel));
    return vreg;
}

template <typename T> struct Mma44;
template <> struct Mma44<f16> {
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 run(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return (float)__builtin_bit_cast(f16, h); }
    static __device__ __forceinline__ unsigned short from_f32(float v) { return __builtin_bit_cast(unsigned short, (f16)v); }
};
template <> struct Mma44<bf16> {
    typedef short v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 run(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return as_f32((unsigned)h << 16); }
    static __device__ __forceinline__ unsigned short from_f32(float v) { return __builtin_bit_cast(unsigned short, (bf16)v); }
};

// ---- the kernel ----------------------------------------------------------------------------------------------------------------
// Work units.  A GROUP STEP is one quantisation group (gr packed rows) of one panel over the workgroup's chunk range = ipg DMA instructions.
// Step ids: 0 .. 2 GA - 1 = (group id >> 1) of gate (even) / up (odd), then 2 GA .. 2 GA + GB - 1 = the groups of down.  Waves draw steps from
// ONE counter in LDS (the hardware balances nothing inside a workgroup: with a static split the slowest wave held the others for 2.6 us at the
// first barrier); a wave's ring therefore runs straight from its last gate/up steps into rows of `down`.  Instruction ids in the per-wave FIFO:
// (step << 8) | k.
template <typename T, int NS>
__global__ void __launch_bounds__(MLP_W * 64, 4) mlp_ring_kernel(MlpParams p) {
    constexpr bool BF = std::is_same_v<T, bf16>;
    unsigned m_lo, m_hi, magic;
    asm("s_mov_b32 %0, 0x000f000f" : "=s"(m_lo));
    asm("s_mov_b32 %0, 0x00f000f0" : "=s"(m_hi));
    asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t_entry = __builtin_readcyclecounter();
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned lds0 = lds_addr_of(smem);
    const unsigned ring_lds = lds0 + (unsigned)wave * (NS * 1024);
    const char* const ring_lane = smem + (size_t)wave * (NS * 1024) + lane * 16;
    const int GA = p.K >> (3 + p.gshift), GB = p.I >> (3 + p.gshift);
    const int nstepsA = 2 * GA, nsteps = nstepsA + GB;
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? 15u : 31u;

    // ---- this workgroup's chunk ranges and this lane's place in a DMA instruction (no integer division: the split comes from the host,
    //      lane / cw through a float reciprocal -- exact for lane < 64, cw <= 16)
    const int cA0 = w * p.qA + (w < p.rA ? w : p.rA), cwA = p.qA + (w < p.rA ? 1 : 0);
    const int cB0 = w * p.qB + (w < p.rB ? w : p.rB), cwB = p.qB + (w < p.rB ? 1 : 0);
    const int rpiA = 1 << p.lrpiA, rpiB = 1 << p.lrpiB;
    const int ipgA = 1 << (p.gshift - p.lrpiA), ipgB = 1 << (p.gshift - p.lrpiB);     // DMA instructions per group step
    int phA = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)cwA));
    const int cA = lane - phA * cwA;
    const bool actA = phA < rpiA;
    phA = actA ? phA : 0;                                               // idle lanes re-read row phase 0 (same cache lines), their sums are dropped
    int phB = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)cwB));
    const int cB = lane - phB * cwB;
    const bool actB = phB < rpiB;
    phB = actB ? phB : 0;
    // per-lane source of (row phase, chunk) in row 0 of each panel; group g starts g << gshift rows further
    const char* const srcG = (const char*)p.gate.qweight + ((size_t)phA * p.I + 4 * (cA0 + cA)) * 4;
    const char* const srcU = (const char*)p.up.qweight + ((size_t)phA * p.I + 4 * (cA0 + cA)) * 4;
    const char* const srcD = (const char*)p.down.qweight + ((size_t)phB * p.N + 4 * (cB0 + cB)) * 4;
    const size_t strideA = (size_t)p.I << (2 + p.lrpiA), strideB = (size_t)p.N << (2 + p.lrpiB);     // bytes per DMA instruction's rows
    const size_t gstrideA = (size_t)p.I << (2 + p.gshift), gstrideB = (size_t)p.N << (2 + p.gshift);   // bytes per group

    // ---- prologue: the small L2-resident pieces first (they return first), then the ring's first NS slots ---------------------
    // DMA jobs, dealt round-robin to the waves.  Tables in LDS are padded to power-of-two rows so a job decodes its (group, dword) with shifts:
    //   x (K * 2 bytes) | scales of gate, up: [GA][1 << lspA] dwords each | zero-point words of gate, up: [GA][1 << lszA] | the same two for down
    char* const cst = smem + p.off_cst;
    const int sSA = GA << p.lspA, sZA = GA << p.lszA, sSB = GB << p.lspB, sZB = GB << p.lszB;      // table sizes in dwords
    const int oSA = 0, oZA = 2 * sSA, oSB = oZA + 2 * sZA, oZB = oSB + sSB;                        // table offsets in dwords
    const int zA0 = (4 * cA0) >> 3, zB0 = (4 * cB0) >> 3;
    {
        const int jX = (p.K * 2 + 1023) >> 10;
        const int jSA = (sSA + 63) >> 6, jZA = (sZA + 63) >> 6, jSB = (sSB + 63) >> 6, jZB = (sZB + 63) >> 6;
        const int e0 = jX, e1 = e0 + 2 * jSA, e2 = e1 + 2 * jZA, e3 = e2 + jSB, e4 = e3 + jZB;
        for (int job = wave; job < e4; job += MLP_W) {
            if (job < e0) {
                const int b = job * 1024 + lane * 16;
                if (b < p.K * 2) dma16_((const char*)p.x + b, lds0 + p.off_x + job * 1024);
            } else if (job < e1) {                                      // scales of gate / up: dword d of group g = columns 4 cA0 + 2 d, + 1
                const int t2 = job - e0, pan = t2 >= jSA, t = t2 - pan * jSA;
                const int e = t * 64 + lane, g = e >> p.lspA, d = e & ((1 << p.lspA) - 1);
                if (g < GA && d < 2 * cwA)
                    dma4_((const char*)(pan ? p.up.scales : p.gate.scales) + ((size_t)g * p.I + 4 * cA0) * 2 + d * 4, lds0 + p.off_cst + (oSA + pan * sSA + t * 64) * 4);
            } else if (job < e2) {
                const int t2 = job - e1, pan = t2 >= jZA, t = t2 - pan * jZA;
                const int e = t * 64 + lane, g = e >> p.lszA, d = e & ((1 << p.lszA) - 1);
                if (g < GA && zA0 + d < (p.I >> 3))
                    dma4_((pan ? p.up.qzeros : p.gate.qzeros) + (size_t)g * (p.I >> 3) + zA0 + d, lds0 + p.off_cst + (oZA + pan * sZA + t * 64) * 4);
            } else if (job < e3) {
                const int t = job - e2;
                const int e = t * 64 + lane, g = e >> p.lspB, d = e & ((1 << p.lspB) - 1);
                if (g < GB && d < 2 * cwB)
                    dma4_((const char*)p.down.scales + ((size_t)g * p.N + 4 * cB0) * 2 + d * 4, lds0 + p.off_cst + (oSB + t * 64) * 4);
            } else {
                const int t = job - e3;
                const int e = t * 64 + lane, g = e >> p.lszB, d = e & ((1 << p.lszB) - 1);
                if (g < GB && zB0 + d < (p.N >> 3))
                    dma4_(p.down.qzeros + (size_t)g * (p.N >> 3) + zB0 + d, lds0 + p.off_cst + (oZB + t * 64) * 4);
            }
        }
    }
    // ---- the step counter: the first MLP_W steps are dealt by wave number (nobody has to wait for an initialised counter), the rest is drawn
    unsigned* const ctr = (unsigned*)(smem + p.off_ctr);
    if (tid == 0) *ctr = MLP_W;                                         // visible after the prologue barrier; nobody draws before it
    // refill cursor: the step being issued, its next instruction, the source of that instruction
    int fstep = wave, fk = 0, fipg = 0;
    unsigned fnext = 0;                                                 // the step drawn ahead (raw atomic result: lane 0 holds it; read when needed)
    const char* fp = nullptr;
    size_t fstride = 0;
    bool exhausted = false;
    int cnt = 0;                                                        // instructions in flight or landed, not yet consumed
    int fifo = 0;                                                       // lane s: the instruction id held by ring slot s
    auto open_step = [&](int st) __attribute__((always_inline)) {      // wave-uniform
        fstep = st; fk = 0;
        if (st >= nsteps) { exhausted = true; return; }
        if (st < nstepsA) { fipg = ipgA; fstride = strideA; fp = ((st & 1) ? srcU : srcG) + (size_t)(st >> 1) * gstrideA; }
        else { fipg = ipgB; fstride = strideB; fp = srcD + (size_t)(st - nstepsA) * gstrideB; }
    };
    auto draw = [&]() __attribute__((always_inline)) -> unsigned {     // one LDS atomic per step, its value is read a whole step later
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;
    };
    auto refill = [&](int slot_off) __attribute__((always_inline)) {   // issue the next instruction of the wave's sequence into this ring slot
        if (exhausted) return;
        dma16_nt_(fp, ring_lds + (unsigned)slot_off);
        fifo = writelane((fstep << 8) | fk, slot_off >> 10, fifo);
        fp += fstride;
        ++cnt;
        if (++fk == fipg) { open_step(uni((int)fnext)); if (!exhausted) fnext = draw(); }
    };
    open_step(wave);
    // the ring's first NS slots come from the wave's dealt step and, if that is shorter than the ring, from steps drawn after the barrier --
    // so the prologue issues at most one step here and tops the ring up right after the barrier
    int first = 0;
    {
        const int n0 = fipg < NS ? fipg : NS;
        for (; first < n0; ++first) {
            dma16_nt_(fp, ring_lds + (unsigned)(first * 1024));
            fifo = writelane((fstep << 8) | fk, first, fifo);
            fp += fstride; ++cnt; ++fk;
        }
    }
    // launch epoch (bumped by the last workgroup of the previous launch on this workspace; nobody can bump it again before every workgroup has
    // published, i.e. long after this read).  A SCALAR load with its own wait, placed here on purpose: a vector load would make hipcc protect its
    // first use -- at the hand-off -- with s_waitcnt vmcnt(0), i.e. behind every prefetched slot of `down`; here its latency hides under the DMAs.
    unsigned epoch;
    unsigned long long dbgp;                                            // header tail words [4:5]: optional timeline buffer (tools/mlplab), 0 = off
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(epoch), "=&s"(dbgp) : "s"(p.hdr) : "memory");
    const unsigned tag = 0x7FE00000u | ((epoch + 1u) & 0x1FFFFFu);
    unsigned long long* const dbg = (unsigned long long*)dbgp;
    auto stamp = [&](int k) __attribute__((always_inline)) {           // per-wave s_memtime stamp k (lab only: one scalar branch when off)
        if (dbg) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (lane == 0) dbg[((size_t)w * MLP_W + wave) * 16 + k] = t;
        }
    };
    if (dbg && lane == 0) dbg[((size_t)w * MLP_W + wave) * 16 + 0] = t_entry;
    stamp(1);
    if (first == NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");       // everything older than the ring's DMAs has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                    // x, every constant table and the step counter are visible
    stamp(2);
    fnext = draw();
    if (fk == fipg) { open_step(uni((int)fnext)); if (!exhausted) fnext = draw(); }
    for (; first < NS; ++first) refill(first * 1024);                   // top the ring up (steps shorter than the ring)

    // ---- the per-instruction math ----------------------------------------------------------------------------------------------
    f32x4 accG[4], accU[4], accg[4];
    float sc[4];
    f16x2 c1[4], c2[4];
    const f16x2 k960 = {(f16)960.f, (f16)960.f};
    const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
#pragma unroll
    for (int c = 0; c < 4; ++c) { accG[c] = f32x4{0.f, 0.f, 0.f, 0.f}; accU[c] = f32x4{0.f, 0.f, 0.f, 0.f}; accg[c] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[c] = 0.f; }
    auto fold = [&](f32x4 (&acc)[4]) __attribute__((always_inline)) {  // close a group: acc += scale * (fp32 sums of the group)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = fmaf(sc[c], accg[c][r], acc[c][r]);
            accg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto set_consts = [&](u32x2 sraw, unsigned zw) __attribute__((always_inline)) {     // zw: the lane's 4 zero-point nibbles in bits 0..15
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned z = (((zw >> (4 * c)) & 15u) + 1u) & zmask;
            c1[c] = as_f16x2(z * 0x00010001u + 0xE400E400u);            // -(1024 + z)
            c2[c] = c1[c] + k960;                                       // -(64 + z)
            const unsigned sw = sraw[c >> 1];
            sc[c] = Mma44<T>::to_f32((unsigned short)((c & 1) ? (sw >> 16) : (sw & 0xffffu)));
        }
    };
    // x given in natural order (x0,x1)(x2,x3)(x4,x5)(x6,x7) -> slot order (k0,k4,k1,k5) (k2,k6,k3,k7), then 8 MFMAs on the lane's 4 columns
    auto mac = [&](const u32x4 qv, const u32x4 t) __attribute__((always_inline)) {
        const u32x2 a01 = {__builtin_amdgcn_perm(t[2], t[0], 0x05040100u), __builtin_amdgcn_perm(t[2], t[0], 0x07060302u)};
        const u32x2 a23 = {__builtin_amdgcn_perm(t[3], t[1], 0x05040100u), __builtin_amdgcn_perm(t[3], t[1], 0x07060302u)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned qw = qv[c], q8 = qw >> 8;
            const f16x2 h0 = as_f16x2((qw & m_lo) | magic) + c1[c];             // k0,k4
            const f16x2 h1 = as_f16x2((qw & m_hi) | magic) * r16 + c2[c];       // k1,k5
            const f16x2 h2 = as_f16x2((q8 & m_lo) | magic) + c1[c];             // k2,k6
            const f16x2 h3 = as_f16x2((q8 & m_hi) | magic) * r16 + c2[c];       // k3,k7
            u32x2 b01, b23;
            if constexpr (BF) {
                auto to_bf = [&](f16x2 hv) __attribute__((always_inline)) -> unsigned {
                    const bf16x2 o = {(bf16)(float)hv[0], (bf16)(float)hv[1]};
                    return __builtin_bit_cast(unsigned, o);
                };
                b01 = u32x2{to_bf(h0), to_bf(h1)};
                b23 = u32x2{to_bf(h2), to_bf(h3)};
            } else {
                b01 = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                b23 = u32x2{__builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
            }
            accg[c] = Mma44<T>::run(a01, b01, accg[c]);
            accg[c] = Mma44<T>::run(a23, b23, accg[c]);
        }
    };
    // the lane's dot products = the block diagonal (register lane & 3), summed over the row phases of its chunk; lanes < cw get the sums
    auto panel_sums = [&](const f32x4 (&acc)[4], bool active, int rpi, int cw, float (&s)[4]) __attribute__((always_inline)) {
        const int d = lane & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = d == 0 ? acc[c][0] : (d == 1 ? acc[c][1] : (d == 2 ? acc[c][2] : acc[c][3]));
            if (!active) v = 0.f;
            for (int off = rpi >> 1; off >= 1; off >>= 1) {              // wave-uniform trip count
                const int src = lane + off * cw;
                const float o = as_f32((unsigned)__builtin_amdgcn_ds_bpermute((src & 63) << 2, (int)as_u32(v)));
                v += (src < 64) ? o : 0.f;
            }
            s[c] = v;
        }
    };

    const unsigned* const cstw = (const unsigned*)cst;
    int slot_off = 0;                                                   // ring slot of the next instruction to consume (bytes)
    float* const red = (float*)(smem + p.off_red);                      // [2][W][CWMAX][4]
    float* const red2 = (float*)(smem + p.off_x);                       // [W][CWMAX][4], after [B1]
    // ================================ phase A: this wave's gate / up steps =========================================================
    {
        const char* const xs = smem + p.off_x;
        const int zselA = ((4 * (cA0 + cA)) >> 3) - zA0, zshA = ((4 * (cA0 + cA)) & 7) * 4;
        int sprev = -1;                                                 // step of the previous instruction (its group is still open in accg)
        while (cnt > 0) {
            const int id = __builtin_amdgcn_readlane(fifo, slot_off >> 10);
            const int st = id >> 8, k = id & 255;
            if (st >= nstepsA) break;                                   // the ring has run on into `down`: phase A is over for this wave
            if (st != sprev) {
                if (sprev >= 0) { if (sprev & 1) fold(accU); else fold(accG); }
                const int pan = st & 1, g = st >> 1;
                const u32x2 sraw = *(const u32x2*)(cstw + oSA + pan * sSA + (g << p.lspA) + 2 * cA);
                const unsigned zw = cstw[oZA + pan * sZA + (g << p.lszA) + zselA] >> zshA;
                set_consts(sraw, zw);
                sprev = st;
            }
            if (exhausted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");
            const u32x4 qv = *(const u32x4*)(ring_lane + slot_off);
            const int row = (((st >> 1) << (p.gshift - p.lrpiA)) + k) << p.lrpiA;
            const u32x4 t = *(const u32x4*)(xs + (size_t)(row + phA) * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot has been read: it may be refilled
            --cnt;
            refill(slot_off);
            mac(qv, t);
            slot_off = (slot_off + 1024 == NS * 1024) ? 0 : slot_off + 1024;
        }
        if (sprev >= 0) { if (sprev & 1) fold(accU); else fold(accG); }
        stamp(3);
        float s[4];
        panel_sums(accG, actA, rpiA, cwA, s);
        if (lane < cwA) *(f32x4*)(red + ((0 * MLP_W + wave) * MLP_CWMAX + lane) * 4) = f32x4{s[0], s[1], s[2], s[3]};
        panel_sums(accU, actA, rpiA, cwA, s);
        if (lane < cwA) *(f32x4*)(red + ((1 * MLP_W + wave) * MLP_CWMAX + lane) * 4) = f32x4{s[0], s[1], s[2], s[3]};
    }
    stamp(4);
    __syncthreads();                                                    // [B1] every wave's gate / up sums are parked
    stamp(5);
    if (tid < 4 * cwA) {                                                // one thread per column of the slice (cw <= 16: inside wave 0)
        const int cc = tid >> 2, q4 = tid & 3;
        float g = 0.f, u = 0.f;
#pragma unroll
        for (int ww = 0; ww < MLP_W; ++ww) {
            g += red[((0 * MLP_W + ww) * MLP_CWMAX + cc) * 4 + q4];
            u += red[((1 * MLP_W + ww) * MLP_CWMAX + cc) * 4 + q4];
        }
        const int n = 4 * (cA0 + cc) + q4;
        if (p.gate.bias) g += Mma44<T>::to_f32(((const unsigned short*)p.gate.bias)[n]);
        if (p.up.bias) u += Mma44<T>::to_f32(((const unsigned short*)p.up.bias)[n]);
        const float a = g / (1.f + __expf(-g)) * u;                     // silu(g) * u on the fp32 sums, one rounding
        const unsigned h = Mma44<T>::from_f32(a);
        const unsigned hn = (unsigned)__shfl_down((int)h, 1, 64);
        if (!(q4 & 1)) {
            const unsigned long long gr8 = (unsigned long long)(h | (hn << 16)) | ((unsigned long long)tag << 32);
            __hip_atomic_store(p.gran + (n >> 1), gr8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // one 8-byte write-through store
        }
    }
    stamp(6);

    // ================================ phase B: this wave's rows of down ==============================================================
    // No workgroup-wide activation: a wave needs the 8 * rpiB activation values per DMA instruction it holds -- 4 rpiB granules, polled by
    // lanes 0 .. 4 rpiB - 1 for every instruction in its ring at once and validated by their tag; lane (row phase ph) then pulls the four
    // granules of its row from lanes 4 ph .. 4 ph + 3 with ds_bpermute.  No barrier: a wave starts as soon as ITS producers have published.
#pragma unroll
    for (int c = 0; c < 4; ++c) accG[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const int zselB = ((4 * (cB0 + cB)) >> 3) - zB0, zshB = ((4 * (cB0 + cB)) & 7) * 4;
        const int npoll = 4 << p.lrpiB;                                 // granules per instruction
        int sprev = -1;
        unsigned spins = 0;
        while (cnt > 0) {
            const int nb = cnt < NS ? cnt : NS;                         // everything in the ring now belongs to `down`
            unsigned long long v[NS];
            unsigned pending = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (s < nb && lane < npoll) pending |= 1u << s;
            int ids[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                int so = slot_off + s * 1024;
                so = so >= NS * 1024 ? so - NS * 1024 : so;
                ids[s] = __builtin_amdgcn_readlane(fifo, so >> 10);
            }
            while (true) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (pending & (1u << s)) {
                        const int row0 = ((((ids[s] >> 8) - nstepsA) << (p.gshift - p.lrpiB)) + (ids[s] & 255)) << p.lrpiB;
                        v[s] = __hip_atomic_load(p.gran + (size_t)row0 * 4 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if ((pending & (1u << s)) && (unsigned)(v[s] >> 32) == tag) pending &= ~(1u << s);
                if (__builtin_amdgcn_ballot_w64(pending != 0) == 0ull) break;
                if (++spins > p.max_spins) {
                    if (lane == 0) __hip_atomic_store(p.hdr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            // every DMA issued before the polls has landed (loads return in order and the polls have returned).
            // One runtime loop over the batch (not NS unrolled copies: this code runs once per wave, straight out of a cold instruction
            // cache): the batch's granule registers move down one place per instruction.
            for (int s = 0; s < nb; ++s) {
                const int id = __builtin_amdgcn_readlane(fifo, slot_off >> 10);
                const int st = id >> 8;
                if (st != sprev) {
                    if (sprev >= 0) fold(accG);
                    const int g = st - nstepsA;
                    const u32x2 sraw = *(const u32x2*)(cstw + oSB + (g << p.lspB) + 2 * cB);
                    const unsigned zw = cstw[oZB + (g << p.lszB) + zselB] >> zshB;
                    set_consts(sraw, zw);
                    sprev = st;
                }
                const u32x4 qv = *(const u32x4*)(ring_lane + slot_off);
                u32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[j] = (unsigned)__builtin_amdgcn_ds_bpermute((4 * phB + j) << 2, (int)(unsigned)(v[0] & 0xffffffffu));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                --cnt;
                refill(slot_off);
                mac(qv, t);
                slot_off = (slot_off + 1024 == NS * 1024) ? 0 : slot_off + 1024;
#pragma unroll
                for (int r = 0; r + 1 < NS; ++r) v[r] = v[r + 1];
            }
            if (cnt > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // a second batch: its DMAs were issued during the first
        }
        if (sprev >= 0) fold(accG);
        stamp(10);
        float s[4];
        panel_sums(accG, actB, rpiB, cwB, s);
        // (parked in the x region -- dead since [B1] -- not in `red`: wave 0 may still be summing the gate / up slabs there)
        if (lane < cwB) *(f32x4*)(red2 + (wave * MLP_CWMAX + lane) * 4) = f32x4{s[0], s[1], s[2], s[3]};
    }
    __syncthreads();                                                    // [B3]
    if (tid < 4 * cwB) {
        const int cc = tid >> 2, q4 = tid & 3;
        float y = 0.f;
#pragma unroll
        for (int ww = 0; ww < MLP_W; ++ww) y += red2[(ww * MLP_CWMAX + cc) * 4 + q4];
        const int n = 4 * (cB0 + cc) + q4;
        if (p.down.bias) y += Mma44<T>::to_f32(((const unsigned short*)p.down.bias)[n]);
        ((unsigned short*)p.out)[n] = Mma44<T>::from_f32(y);
    }
    stamp(11);
    if (tid == 0) {                                                     // every wave of this workgroup has consumed its granules: arrive
        const unsigned arrived = __hip_atomic_fetch_add(p.hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (unsigned)p.nwg - 1u) {                          // last one: every granule of this epoch has been consumed everywhere
            __hip_atomic_store(p.hdr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.hdr, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) silu_mul2_kernel(const T* __restrict__ g, const T* __restrict__ u, T* __restrict__ out, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float a = DType<T>::to_f32(g[i]), b = DType<T>::to_f32(u[i]);
        out[i] = DType<T>::from_f32(a / (1.f + __expf(-a)) * b);
    }
}

int g_cu_count[64] = {0};          // per device ordinal, filled by init_mlp_device (gptq_init); 0 = not initialised

constexpr int mlp_ns_options[] = {8, 7, 6, 4};

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
    auto grant = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
    grant((const void*)mlp_ring_kernel<f16, 8>); grant((const void*)mlp_ring_kernel<f16, 7>); grant((const void*)mlp_ring_kernel<f16, 6>); grant((const void*)mlp_ring_kernel<f16, 4>);
    grant((const void*)mlp_ring_kernel<bf16, 8>); grant((const void*)mlp_ring_kernel<bf16, 7>); grant((const void*)mlp_ring_kernel<bf16, 6>); grant((const void*)mlp_ring_kernel<bf16, 4>);
    return e;
}

static bool mlp_layer_ok(const gptq_layer_t& L) {
    const int gr = L.group_size / 8;
    return L.bits == 4 && (L.dtype == GPTQ_F16 || L.dtype == GPTQ_BF16) && L.g_idx == nullptr && L.epilogue == GPTQ_EPI_NONE &&
           L.group_size % 8 == 0 && gr >= 4 && (gr & (gr - 1)) == 0 && L.K % L.group_size == 0;
}

// nwg_override > 0: tests / the lab run the same kernel on fewer workgroups (every one of them must still be resident at once)
MlpPlan plan_mlp(const gptq_layer_t& gate, const gptq_layer_t& up, const gptq_layer_t& down, int M, int nwg_override) {
    MlpPlan pl{};
    if (M != 1) return pl;
    if (!mlp_layer_ok(gate) || !mlp_layer_ok(up) || !mlp_layer_ok(down)) return pl;
    if (gate.K != up.K || gate.N != up.N || down.K != gate.N) return pl;
    if (gate.group_size != up.group_size || gate.group_size != down.group_size) return pl;
    if (gate.dtype != up.dtype || gate.dtype != down.dtype || gate.zero_mode != up.zero_mode || gate.zero_mode != down.zero_mode) return pl;
    int nwg = nwg_override;
    if (nwg <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return pl;
        nwg = g_cu_count[dev];
    }
    if (nwg <= 0) return pl;
    const int K = gate.K, I = gate.N, N = down.N;
    if (I > 16384 || K > 8192 || K % 8 || I % 8) return pl;
    const int chA = I / 4, chB = N / 4;
    if (chA < nwg || chB < nwg) return pl;
    const int cwA = (chA + nwg - 1) / nwg, cwB = (chB + nwg - 1) / nwg;
    if (cwA > MLP_CWMAX || cwB > MLP_CWMAX) return pl;
    const int gr = gate.group_size / 8;
    const int GA = K / gate.group_size, GB = I / gate.group_size;
    if (2 * GA < MLP_W) return pl;                                      // the first 16 steps are dealt by wave number
    auto lg2f = [](int v) { int r = 0; while ((2 << r) <= v) ++r; return r; };      // floor(log2 v)
    auto lg2c = [](int v) { int r = 0; while ((1 << r) < v) ++r; return r; };       // ceil(log2 v)
    pl.gshift = lg2f(gr);
    // rows per DMA instruction: what the WIDEST workgroup panel leaves of 64 lanes, a power of two that divides the rows of a group
    pl.lrpiA = std::min(lg2f(64 / cwA), pl.gshift);
    pl.lrpiB = std::min(lg2f(64 / cwB), pl.gshift);
    if ((gr >> pl.lrpiA) > 255 || (gr >> pl.lrpiB) > 255) return pl;
    pl.qA = chA / nwg; pl.rA = chA % nwg; pl.qB = chB / nwg; pl.rB = chB % nwg;
    // constant tables in LDS, rows padded to powers of two: scales 2 cw dwords per group row, zero-point words covering 4 cw nibbles at any alignment
    pl.lspA = lg2c(2 * cwA); pl.lszA = lg2c((4 * cwA + 7) / 8 + 1);
    pl.lspB = lg2c(2 * cwB); pl.lszB = lg2c((4 * cwB + 7) / 8 + 1);
    const size_t cst = ((size_t)2 * ((size_t)GA << pl.lspA) + (size_t)2 * ((size_t)GA << pl.lszA) + ((size_t)GB << pl.lspB) + ((size_t)GB << pl.lszB)) * 4 + 256;
    const size_t xb = std::max(((size_t)K * 2 + 1023) / 1024 * 1024, (size_t)MLP_W * MLP_CWMAX * 4 * sizeof(float));    // x, later the phase-B slabs
    const size_t red = (size_t)2 * MLP_W * MLP_CWMAX * 4 * sizeof(float);
    for (int ns : mlp_ns_options) {
        const size_t ring = (size_t)MLP_W * ns * 1024;
        size_t off = ring;
        const size_t ox = off; off += xb;
        const size_t oc = off; off += (cst + 15) / 16 * 16;
        const size_t orr = off; off += red;
        const size_t octr = off; off += 16;
        if (off <= 160 * 1024 && off > 80 * 1024) {                      // > half the LDS: at most one workgroup per CU, whatever else is resident
            pl.ok = true; pl.ns = ns; pl.nwg = nwg;
            pl.off_x = (int)ox; pl.off_cst = (int)oc; pl.off_red = (int)orr; pl.off_ctr = (int)octr;
            pl.lds_bytes = off;
            pl.exchange_bytes = ((size_t)I / 2 * 8 + 255) / 256 * 256;
            return pl;
        }
    }
    return pl;
}

hipError_t launch_mlp(const gptq_layer_t& gate, const gptq_layer_t& up, const gptq_layer_t& down, const MlpPlan& pl, const void* x, void* out,
                      void* ws_header, void* exchange, hipStream_t st) {
    if (!pl.ok) return hipErrorInvalidValue;
    MlpParams p{};
    p.gate = MlpLayerArgs{gate.qweight, gate.qzeros, gate.scales, gate.bias};
    p.up = MlpLayerArgs{up.qweight, up.qzeros, up.scales, up.bias};
    p.down = MlpLayerArgs{down.qweight, down.qzeros, down.scales, down.bias};
    p.x = x; p.out = out;
    p.hdr = (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES);
    p.gran = (unsigned long long*)exchange;
    p.K = gate.K; p.I = gate.N; p.N = down.N;
    p.gshift = pl.gshift;
    p.zero_mode = gate.zero_mode;
    p.nwg = pl.nwg;
    p.max_spins = 200000u;                                              // x (s_sleep 4 + an L2 round trip) ~ 0.2 s: a stuck peer ends the wait, never the queue
    p.qA = pl.qA; p.rA = pl.rA; p.qB = pl.qB; p.rB = pl.rB;
    p.lrpiA = pl.lrpiA; p.lrpiB = pl.lrpiB;
    p.lspA = pl.lspA; p.lszA = pl.lszA; p.lspB = pl.lspB; p.lszB = pl.lszB;
    p.off_x = pl.off_x; p.off_cst = pl.off_cst; p.off_red = pl.off_red; p.off_ctr = pl.off_ctr;
    const bool bf = gate.dtype == GPTQ_BF16;
#define MLP_LAUNCH(NSV)                                                                                                               \
    case NSV:                                                                                                                         \
        if (bf) hipLaunchKernelGGL((mlp_ring_kernel<bf16, NSV>), dim3(pl.nwg), dim3(MLP_W * 64), pl.lds_bytes, st, p);                \
        else hipLaunchKernelGGL((mlp_ring_kernel<f16, NSV>), dim3(pl.nwg), dim3(MLP_W * 64), pl.lds_bytes, st, p);                    \
        break;
    switch (pl.ns) {
        MLP_LAUNCH(8) MLP_LAUNCH(7) MLP_LAUNCH(6) MLP_LAUNCH(4)
        default: return hipErrorInvalidValue;
    }
#undef MLP_LAUNCH
    return hipGetLastError();
}

}  // namespace gptq
