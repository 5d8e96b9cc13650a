// gemm_mid.hip -- 4-bit fp16 / bf16 QuantLinear forward for 17 .. 128 rows of x ("batched decode" and small prefill chunks).
//
// Role in the reference: the regime between the one-row kernels and the dequantise-then-GEMM fallbacks -- exllamav2's MAX_Q_GEMM_ROWS = 50
// (autogptq_extension/exllamav2/cuda/q_gemm.cu:118, config.h:4), qlinear_cuda.py:34,212 (kernel_switch_threshold) and the batch > 1 rows of
// the reference's own benchmark table.  Nothing is derived from those kernels.
//
// Why another kernel: gemm_stream64_kernel (gemm.hip) brings x to the matrix core through registers -- 16-byte loads of 64-byte row
// segments from L2 per (K-step, row tile), then 4 ds_bpermute to reach the MFMA's lane layout -- and that path, not the weights, is what
// its time grows with beyond 32 rows (profiles/r03_stream64_ablation.log: 22.8 us at M = 64 on 4096 x 11008, 18.0 without the x loads; the
// weights alone stream in 8.6).  Here NOTHING a wave consumes passes through a VGPR before it is used:
//   * weights: one LDS DMA (global_load_lds_dwordx4, 1 KiB, nontemporal) per 32-deep K-step of the 64-column strip, as in stream64;
//   * x: one LDS DMA per (K-step, 16-row tile) -- lanes 4q .. 4q+3 fetch the 64 contiguous bytes of row q (coalesced), their four 16-byte
//     octets rotated by q >> 2 so that the MFMA's lane (row i, k-octet kg) reads slot 4 i + ((kg + (i >> 2)) & 3) with ONE conflict-free
//     ds_read_b128 (the 16 lanes of a quarter-wave hit 16 different 16-byte bank groups);
//   * group constants: the (scales, zero-point) rows of all the groups a wave will touch are DMA'd once, up front, into a per-wave LDS
//     table (256 B per group) -- the K loop's only VMEM instructions are its own DMAs, so `s_waitcnt vmcnt(newer stages x (1 + RT))` is exact;
//   * D stages of (1 + RT) KiB per wave are in flight; waves split the workgroup's K range (no barrier in the K loop); addresses are
//     scalar base (+ K-step) + a fixed 32-bit lane offset: no vector address arithmetic per DMA.
// K slices of a strip are combined inside the launch: slices 1.. write their fp32 partial tile with 16-byte write-through (sc1) stores,
// drain them and set ONE flag word each; slice 0 (the owner) polls the flags (bounded), adds the partials in slice order (bit-reproducible)
// and clears the flags.  Flags live in the zeroed ticket half of the workspace header (gptq_mi355x.h): zero before and after every launch.
#include <type_traits>
#include <utility>

#include "common.cuh"
#include "launch.h"

namespace gptq {

namespace midk {

struct MidSeg {
    const unsigned* qweight;
    const unsigned* qzeros;
    const void* scales;
    const void* bias;
    void* out;
    int N;         // columns of this layer (multiple of 64)
    int blk_end;   // cumulative strip count up to and including this layer
    int col0;      // first column of this layer in the concatenated partial slab
    int pad_;
};
struct MidParams {
    MidSeg seg[4];       // up to four layers that read the same x: the grid runs over all their strips
    const void* x;
    float* partial;      // [ksplit - 1][M][nsum] fp32 partial tiles of K slices 1 .. ksplit - 1
    unsigned* flags;     // workspace header, ticket half: word [strip * 8 + slice], zero before and after every launch (flag combine)
    unsigned* epochs;    // workspace header, second half: per-strip launch epoch (granule combine; monotonic, bumped by the owner slice, never reset)
    unsigned* err;       // sticky error word (header tail): a bounded wait gave up
    int gran;            // 1: K slices 1.. publish {fp32, tag} granules (8 bytes per value) that the owner validates itself -- ONE memory hop;
                         // 0: fp32 partial tiles + one flag word per slice (publish, drain, flag, read: three)
    int nseg, M, K, zero_mode, ksplit, ksteps_per_split, nsum;
    int strips_total;    // strips over all layers; the grid is strips_total x row_blocks x ksplit
    int row_blocks;
    int lg_gsteps;       // log2(group_size / 32)
    int tab_bytes;       // per-wave group-constant table (256 B per group the wave can touch)
    unsigned max_spins;
};

__device__ __forceinline__ unsigned and_or(unsigned q, unsigned mask, unsigned magic) { return (q & mask) | magic; }
__device__ __forceinline__ unsigned f16x2_bits(f16x2 v) { return __builtin_bit_cast(unsigned, v); }

// 16 bytes per lane global -> LDS (lds_dst + lane * 16); source = scalar base + 32-bit per-lane byte offset
__device__ __forceinline__ void dma16_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16_sv_nt(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void store16_sc1(void* dst, f32x4 v) {      // write-through: visible to every XCD once drained
    // s_nop inside the string: hipcc pads nothing behind an asm statement, and the data registers are dead to it -- an instruction right behind
    // the store that overwrites them races with the store's own read of them (seen in the tiled kernel's balanced tail, tools/tail_diag.py)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

// one column: (scale bits, zero-point) -> fragment of one 4-bit word = 8 consecutive k in the slot order k0,k4,k1,k5,k2,k6,k3,k7;
// exact w - z in packed fp16 (magic number 0x6400), then x scale: the reference's scales * (w - z), bit for bit
template <typename T> struct Deq1;
template <> struct Deq1<f16> {
    f16x2 s2, c1, c2;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        u32x4 o;
        o[0] = f16x2_bits((as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1) * s2);
        o[1] = f16x2_bits((as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2) * s2);
        o[2] = f16x2_bits((as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1) * s2);
        o[3] = f16x2_bits((as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2) * s2);
        return o;
    }
};
template <> struct Deq1<bf16> {
    f16x2 c1, c2;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sbits);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
        const f16x2 k960 = {(f16)960.f, (f16)960.f};
        c2 = c1 + k960;
    }
    static __device__ __forceinline__ unsigned scaled_pair(f16x2 h, float sc) {      // exact fp32 product, one rounding to bf16
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        float lo, hi;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hb), "v"(sc));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(hb), "v"(sc));
        const bf16x2 v = {(bf16)lo, (bf16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ __forceinline__ u32x4 frag(unsigned q) const {
        const unsigned q8 = q >> 8;
        const f16x2 r16 = {(f16)0.0625f, (f16)0.0625f};
        u32x4 o;
        o[0] = scaled_pair(as_f16x2(and_or(q, 0x000f000fu, 0x64006400u)) + c1, s);
        o[1] = scaled_pair(as_f16x2(and_or(q, 0x00f000f0u, 0x64006400u)) * r16 + c2, s);
        o[2] = scaled_pair(as_f16x2(and_or(q8, 0x000f000fu, 0x64006400u)) + c1, s);
        o[3] = scaled_pair(as_f16x2(and_or(q8, 0x00f000f0u, 0x64006400u)) * r16 + c2, s);
        return o;
    }
};
// 8-bit: a lane's 8 consecutive k of one column are two words (packed rows 2 kg and 2 kg + 1); v_perm joins their halves so that a byte mask pairs
// (f0, f4), (f2, f6) and, shifted by 8, (f1, f5), (f3, f7): the same slot order as the 4-bit fragments, exact w - z with the 0x6400 magic number
// (|w - z| <= 256), then x scale (fp16: packed multiply; bf16: exact fp32 product, one rounding)
template <typename T> struct Deq8;
template <> struct Deq8<f16> {
    f16x2 s2, c1;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
    }
    __device__ __forceinline__ u32x4 frag(unsigned w0, unsigned w1) const {
        const unsigned lo = __builtin_amdgcn_perm(w1, w0, 0x05040100u);      // f0 f1 | f4 f5
        const unsigned hi = __builtin_amdgcn_perm(w1, w0, 0x07060302u);      // f2 f3 | f6 f7
        u32x4 o;
        o[0] = f16x2_bits((as_f16x2(and_or(lo, 0x00ff00ffu, 0x64006400u)) + c1) * s2);
        o[1] = f16x2_bits((as_f16x2(and_or(lo >> 8, 0x00ff00ffu, 0x64006400u)) + c1) * s2);
        o[2] = f16x2_bits((as_f16x2(and_or(hi, 0x00ff00ffu, 0x64006400u)) + c1) * s2);
        o[3] = f16x2_bits((as_f16x2(and_or(hi >> 8, 0x00ff00ffu, 0x64006400u)) + c1) * s2);
        return o;
    }
};
template <> struct Deq8<bf16> {
    f16x2 c1;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sbits);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);
    }
    __device__ __forceinline__ u32x4 frag(unsigned w0, unsigned w1) const {
        const unsigned lo = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
        const unsigned hi = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
        u32x4 o;
        o[0] = Deq1<bf16>::scaled_pair(as_f16x2(and_or(lo, 0x00ff00ffu, 0x64006400u)) + c1, s);
        o[1] = Deq1<bf16>::scaled_pair(as_f16x2(and_or(lo >> 8, 0x00ff00ffu, 0x64006400u)) + c1, s);
        o[2] = Deq1<bf16>::scaled_pair(as_f16x2(and_or(hi, 0x00ff00ffu, 0x64006400u)) + c1, s);
        o[3] = Deq1<bf16>::scaled_pair(as_f16x2(and_or(hi >> 8, 0x00ff00ffu, 0x64006400u)) + c1, s);
        return o;
    }
};
// 3-bit: the lane's 8 consecutive k of one column are a 24-bit window of the column's 96-bit K-step (three packed rows); low half = bits 0..15
// (f0..f4 at 0, 3, 6, 9, 12), high half = bits 12..27 (f4..f7 at 0, 3, 6, 9): pairs (f0, f4), (f1, f5), (f2, f6) at bits 0 / 3 / 6 and (f3, f7) at
// bit 3 of t >> 6 -- again the 4-bit slot order; exact w - z by the magic number with the shifted constants (gemv.hip: MagicF16<3>)
template <typename T> struct Deq3;
template <> struct Deq3<f16> {
    f16x2 s2, c1, c3, c6;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);             // -(1024 + z)
        const f16x2 k896 = {(f16)896.f, (f16)896.f}, k1008 = {(f16)1008.f, (f16)1008.f};
        c3 = c1 + k896;                                           // -(128 + z), exact
        c6 = c1 + k1008;                                          // -(16 + z), exact
    }
    __device__ __forceinline__ void diffs(unsigned v, f16x2 (&h)[4]) const {
        const unsigned t = __builtin_amdgcn_perm(v >> 12, v, 0x05040100u);
        const unsigned t6 = t >> 6;
        const f16x2 r8 = {(f16)0.125f, (f16)0.125f}, r64 = {(f16)0.015625f, (f16)0.015625f};
        h[0] = as_f16x2(and_or(t, 0x00070007u, 0x64006400u)) + c1;             // k0,k4 : w - z
        h[1] = as_f16x2(and_or(t, 0x00380038u, 0x64006400u)) * r8 + c3;        // k1,k5
        h[2] = as_f16x2(and_or(t, 0x01C001C0u, 0x64006400u)) * r64 + c6;       // k2,k6
        h[3] = as_f16x2(and_or(t6, 0x00380038u, 0x64006400u)) * r8 + c3;       // k3,k7
    }
    __device__ __forceinline__ u32x4 frag(unsigned v) const {
        f16x2 h[4];
        diffs(v, h);
        return u32x4{f16x2_bits(h[0] * s2), f16x2_bits(h[1] * s2), f16x2_bits(h[2] * s2), f16x2_bits(h[3] * s2)};
    }
};
template <> struct Deq3<bf16> {
    Deq3<f16> d;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        d.setup(0x3c00u, z);
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sbits);
    }
    __device__ __forceinline__ u32x4 frag(unsigned v) const {
        f16x2 h[4];
        d.diffs(v, h);
        return u32x4{Deq1<bf16>::scaled_pair(h[0], s), Deq1<bf16>::scaled_pair(h[1], s), Deq1<bf16>::scaled_pair(h[2], s), Deq1<bf16>::scaled_pair(h[3], s)};
    }
};
// 2-bit: the lane's 8 consecutive k of one column are 16 bits of a word (16 values per word: half kg & 1 of packed row kg >> 1); v_perm spreads the two
// bytes over the halves of a register (f0..f3 at bits 0, 2, 4, 6 of the low half, f4..f7 of the high half): pairs (f0, f4), (f1, f5), (f2, f6), (f3, f7)
// -- the 4-bit slot order again -- each ONE v_and_or + ONE packed fma by 2^-s with -(1024 * 2^-s + z), exact
template <typename T> struct Deq2;
template <> struct Deq2<f16> {
    f16x2 s2, c1, c2, c4, c6;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        s2 = as_f16x2(sbits * 0x00010001u);
        c1 = as_f16x2(z * 0x00010001u + 0xE400E400u);             // -(1024 + z)
        const f16x2 k768 = {(f16)768.f, (f16)768.f}, k960 = {(f16)960.f, (f16)960.f}, k1008 = {(f16)1008.f, (f16)1008.f};
        c2 = c1 + k768;                                           // -(256 + z), exact
        c4 = c1 + k960;                                           // -(64 + z)
        c6 = c1 + k1008;                                          // -(16 + z)
    }
    __device__ __forceinline__ void diffs(unsigned v16, f16x2 (&h)[4]) const {
        const unsigned t = __builtin_amdgcn_perm(v16, v16, 0x0c010c00u);       // byte 0 -> bits 0..7, byte 1 -> bits 16..23
        const f16x2 r4 = {(f16)0.25f, (f16)0.25f}, r16 = {(f16)0.0625f, (f16)0.0625f}, r64 = {(f16)0.015625f, (f16)0.015625f};
        h[0] = as_f16x2(and_or(t, 0x00030003u, 0x64006400u)) + c1;             // k0,k4 : w - z
        h[1] = as_f16x2(and_or(t, 0x000C000Cu, 0x64006400u)) * r4 + c2;        // k1,k5
        h[2] = as_f16x2(and_or(t, 0x00300030u, 0x64006400u)) * r16 + c4;       // k2,k6
        h[3] = as_f16x2(and_or(t, 0x00C000C0u, 0x64006400u)) * r64 + c6;       // k3,k7
    }
    __device__ __forceinline__ u32x4 frag(unsigned v) const {
        f16x2 h[4];
        diffs(v, h);
        return u32x4{f16x2_bits(h[0] * s2), f16x2_bits(h[1] * s2), f16x2_bits(h[2] * s2), f16x2_bits(h[3] * s2)};
    }
};
template <> struct Deq2<bf16> {
    Deq2<f16> d;
    float s;
    __device__ __forceinline__ void setup(unsigned sbits, unsigned z) {
        d.setup(0x3c00u, z);
        s = (float)__builtin_bit_cast(bf16, (unsigned short)sbits);
    }
    __device__ __forceinline__ u32x4 frag(unsigned v) const {
        f16x2 h[4];
        d.diffs(v, h);
        return u32x4{Deq1<bf16>::scaled_pair(h[0], s), Deq1<bf16>::scaled_pair(h[1], s), Deq1<bf16>::scaled_pair(h[2], s), Deq1<bf16>::scaled_pair(h[3], s)};
    }
};
template <typename T> struct Mma16;
template <> struct Mma16<f16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ u32x2 pack4(f32x4 v) {
    struct { T a, b, c, d; } o{DType<T>::from_f32(v[0]), DType<T>::from_f32(v[1]), DType<T>::from_f32(v[2]), DType<T>::from_f32(v[3])};
    return __builtin_bit_cast(u32x2, o);
}

// RT = row tiles of 16 (M <= 16 RT); D = stages of one K-step in flight per wave.  512 threads: 8 waves split the K range.
// XREG (experiment, tuning.reserved[1] = 1): x fragments by ordinary 16-byte loads (same coalesced lane layout) into registers, written to the
// stage with ds_write_b128 when the stage is consumed -- the L1 path (64 B/clk per CU) instead of the LDS-DMA path (~17 B/clk measured).
// CW = 64-column halves per strip (1: 64-column strips; 2: 128-column strips -- every x fragment feeds 8 MFMAs instead of 4, which halves
// the x bytes a CU pulls from L2 per flop: what bounds this kernel is the ~50 GB/s a CU gets out of the L2 when every CU reads the same x
// (tools/xfetch_lab.hip, profiles/r03_xfetch_lab.log), by DMA or through registers alike).  CW = 2 needs RT <= 4 (128 accumulator registers).
// BITS = 8 (CW = 1): a 32-deep K-step is 8 packed rows = two weight DMAs; the lane reads its two words per column from them (rows 2 kg, 2 kg + 1).
template <typename T, int RT, int D, bool XREG = false, int CW = 1, int BITS = 4>
__global__ void __launch_bounds__(512) gemm_mid_kernel(MidParams p) {
    static_assert(BITS == 4 || ((BITS == 8 || BITS == 3 || BITS == 2) && CW == 1 && !XREG), "2- / 3- / 8-bit: 64-column strips, x by DMA");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WD = BITS == 8 ? 2 : CW;                        // weight DMAs (KiB) per stage
    constexpr int SB = (WD + RT) * 1024;                          // bytes of one stage: [weights WD x 1 KiB][x row tile 0] .. [x row tile RT-1]
    constexpr int OPS = XREG ? WD : WD + RT;                      // asm-issued (compiler-invisible) VMEM instructions per stage; XREG: the x loads are the compiler's
    constexpr int VOPS = WD + RT;                                 // all VMEM instructions per stage
    constexpr int SC = 64 * CW;                                   // columns of a strip
    static_assert((D - 1) * VOPS <= 63, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63, W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, kg = lane >> 4;
    const int wave_bytes = D * SB + p.tab_bytes;
    char* const wbase = smem + (size_t)wave * wave_bytes;         // this wave's stages, then its group-constant table
    const unsigned w_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(wbase));
    // logical block -> (strip over all layers, K slice): slices of one strip are adjacent logical ids (one XCD after the remap)
    // (a workgroup owns 16 RT rows of x: row blocks are what fills the chip on narrow layers WITHOUT K slices and their combine -- the weights of a
    // strip are then streamed once per row block, out of the L2 after the first)
    const int Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = Lb / p.ksplit, ks = Lb - tile * p.ksplit;    // tile = (row block, strip): index of the flag words too
    const int stile = tile / p.row_blocks, rb = tile - stile * p.row_blocks;   // the row blocks of a strip are adjacent ids: same XCD, same moment -> one HBM fetch of its weights
    const int m0 = rb * (RT * 16);
    int sI = 0;
    while (sI + 1 < p.nseg && stile >= p.seg[sI].blk_end) ++sI;   // wave-uniform (kernel arguments only)
    const MidSeg& sg = p.seg[sI];
    const int strip = stile - (sI ? p.seg[sI - 1].blk_end : 0);
    const int N = sg.N;
    const int S = p.K >> 5;
    const int b0 = ks * p.ksteps_per_split, b1 = min(b0 + p.ksteps_per_split, S);
    const int spw = (b1 - b0 + W - 1) / W;
    const int ws = b0 + wave * spw, we = min(ws + spw, b1);
    const unsigned zmask = (p.zero_mode == GPTQ_ZERO_WRAP) ? ((1u << BITS) - 1u) : ((2u << BITS) - 1u);      // stored zero + 1, wrapped to the field or not

#ifdef GPTQ_MID_TL
    // lab build: per-wave s_memtime stamps in the last 1 KiB of the wave's table area (plan_mid adds it), dumped to the buffer whose address the
    // host put into header-tail words 4..5 (tools/mid_timeline.py)
    unsigned long long* const tl_lds = (unsigned long long*)(wbase + wave_bytes - 1024);
    int tl_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0 && tl_n < 126) tl_lds[tl_n] = t;
        ++tl_n;
    };
    stamp();
#else
    auto stamp = [&]() __attribute__((always_inline)) {};
#endif
    f32x4 acc[RT][CW][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int h = 0; h < CW; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[rt][h][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (ws < we) {
        const int g_first = ws >> p.lg_gsteps, g_last = (we - 1) >> p.lg_gsteps;
        {   // group constants of every group this wave touches -> table (256 B per (group, 64-column half): scales of the 64 columns, then the 32 B of zero-points)
            const int sub = lane >> 4, w16 = lane & 15;
            const unsigned tab_lds = w_lds + D * SB;
            const int units = (g_last - g_first + 1) * CW;
            for (int it = 0; it * 4 < units; ++it) {
                const int u = min(it * 4 + sub, units - 1);
                const int gg = g_first + u / CW, c0 = strip * SC + (u % CW) * 64;
                const char* src;
                if constexpr (BITS == 3) {
                    // 64 columns x 3 bits = 24 bytes at byte 24 * strip of the row: not 16-byte aligned for odd strips -- three aligned pieces from
                    // floor16 on cover it; a piece that would start beyond the row is never needed and re-reads the row's last 16 bytes
                    const size_t rowb = (size_t)N * 3 / 8, base16 = ((size_t)strip * 24) & ~(size_t)15;
                    const size_t pz = base16 + (size_t)min(w16 - 8, 2) * 16;
                    src = (w16 < 8) ? (const char*)((const T*)sg.scales + (size_t)gg * N + c0 + w16 * 8)
                                    : (const char*)sg.qzeros + (size_t)gg * rowb + (pz + 16 <= rowb ? pz : rowb - 16);
                } else {
                    constexpr int ZL = BITS / 2;                  // 16-byte pieces of zero-points per 64 columns (2-bit: 16 B, 4-bit: 32 B, 8-bit: 64 B)
                    src = (w16 < 8) ? (const char*)((const T*)sg.scales + (size_t)gg * N + c0 + w16 * 8)
                                    : (const char*)(sg.qzeros + (size_t)gg * (N * BITS / 32) + (c0 * BITS / 32) + min(w16 - 8, ZL - 1) * 4);
                }
                lds_dma16(src, tab_lds + it * 1024);
            }
        }
        // fixed per-lane source offsets (bytes)
        unsigned woff[WD];
#pragma unroll
        for (int h = 0; h < WD; ++h)
            woff[h] = BITS == 8 ? (unsigned)(((size_t)(4 * h + kg) * N + strip * SC + j16 * 4) * 4)       // DMA h: packed rows 4 h .. 4 h + 3
                      : BITS == 3 ? (unsigned)(((size_t)min(kg, 2) * N + strip * SC + j16 * 4) * 4)       // rows 0 .. 2 of the K-step (lanes 48..63 do not take part)
                      : BITS == 2 ? (unsigned)(((size_t)min(kg, 1) * N + strip * SC + j16 * 4) * 4)       // rows 0 .. 1 (lanes 32..63 do not take part)
                                  : (unsigned)(((size_t)kg * N + strip * SC + h * 64 + j16 * 4) * 4);      // DMA h: 64-column half h
        unsigned xoff[RT];
        {
            const int q = lane >> 2, a = lane & 3, oct = (a - (q >> 2)) & 3;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) xoff[rt] = (unsigned)(((size_t)min(m0 + rt * 16 + q, p.M - 1) * p.K + oct * 8) * 2);
        }
        const unsigned a_slot = (unsigned)((4 * j16 + ((kg + (j16 >> 2)) & 3)) * 16);   // where the MFMA lane (row j16, k-octet kg) finds its 16 bytes
        const char* const qw = (const char*)sg.qweight;
        const char* const xb = (const char*)p.x;
        const size_t wstep = (size_t)N * BITS * 4;                // bytes of the packed rows of one K-step (32 k = BITS rows)
        // 3-bit: the lane's 24-bit window [24 kg, 24 kg + 24) of the 96-bit K-step: words (w0, w1) >> 0 / 24 for kg = 0 / 1, (w1, w2) >> 16 / 40 for kg = 2 / 3
        const unsigned w3_sh = kg == 0 ? 0u : (kg == 1 ? 24u : (kg == 2 ? 16u : 40u));
        const int zstart = (int)(((size_t)strip * 24) & 15);      // 3-bit: where the strip's zero-points begin inside the table's zero area (0 or 8)
        const unsigned w8_slot = (unsigned)((kg >> 1) * 1024 + ((((2 * kg) & 3) * 16 + j16) * 16));   // 8-bit: packed row 2 kg of the lane's 4 columns (row 2 kg + 1: + 256)

        u32x4 xr[XREG ? D : 1][XREG ? RT : 1];
        auto issue = [&](int s, int stage) __attribute__((always_inline)) {
            const unsigned dst = w_lds + stage * SB;
            const char* xs = xb + (size_t)s * 64;
            if constexpr (XREG) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) xr[stage][rt] = *(const u32x4*)(xs + xoff[rt]);
            }
            const char* wsrc = qw + (size_t)s * wstep;
#pragma unroll
            for (int h = 0; h < WD; ++h) {
                if constexpr (BITS == 3 || BITS == 2) {
                    if (lane < 16 * BITS) dma16_sv_nt(wsrc, woff[h], dst + h * 1024);   // three / two packed rows = 768 / 512 bytes of the 1 KiB slot
                } else {
                    dma16_sv_nt(wsrc, woff[h], dst + h * 1024);
                }
            }
            if constexpr (!XREG) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) dma16_sv(xs, xoff[rt], dst + (WD + rt) * 1024);
            }
        };
        using DQ = std::conditional_t<BITS == 8, Deq8<T>, std::conditional_t<BITS == 3, Deq3<T>, std::conditional_t<BITS == 2, Deq2<T>, Deq1<T>>>>;
        DQ dq[CW][4];
        int g_cur = -1;
        auto consume = [&](int s, int stage) __attribute__((always_inline)) {
            const char* st = wbase + stage * SB;
            u32x4 qv[BITS == 3 ? 3 : (BITS == 2 ? 2 : WD)];
            if constexpr (BITS == 8) {
                qv[0] = *(const u32x4*)(st + w8_slot);
                qv[1] = *(const u32x4*)(st + w8_slot + 256);
            } else if constexpr (BITS == 3 || BITS == 2) {
#pragma unroll
                for (int r = 0; r < BITS; ++r) qv[r] = *(const u32x4*)(st + r * 256 + j16 * 16);
            } else {
#pragma unroll
                for (int h = 0; h < CW; ++h) qv[h] = *(const u32x4*)(st + h * 1024 + lane * 16);
            }
            const int g = s >> p.lg_gsteps;
            if (g != g_cur) {                                     // wave-uniform
                g_cur = g;
#pragma unroll
                for (int h = 0; h < CW; ++h) {
                    const char* tb = wbase + D * SB + ((g - g_first) * CW + h) * 256;
                    const u32x2 sraw = *(const u32x2*)(tb + j16 * 8);
                    unsigned zz;
                    if constexpr (BITS == 8) zz = *(const unsigned*)(tb + 128 + j16 * 4);
                    else if constexpr (BITS == 3) {               // 12 bits at bit 8 zstart + 12 j16 of the zero area (two aligned words, funnel shift)
                        const unsigned bit = 8u * (unsigned)zstart + 12u * (unsigned)j16;
                        const unsigned* za = (const unsigned*)(tb + 128) + (bit >> 5);
                        zz = (unsigned)((((unsigned long long)za[1] << 32) | za[0]) >> (bit & 31u));
                    } else if constexpr (BITS == 2) zz = *(const unsigned char*)(tb + 128 + j16);
                    else zz = *(const unsigned short*)(tb + 128 + j16 * 2);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const unsigned sw = sraw[t >> 1];
                        const unsigned zf = (zz >> (BITS * t)) & ((1u << BITS) - 1u);
                        dq[h][t].setup((t & 1) ? (sw >> 16) : (sw & 0xffffu), (zf + 1u) & zmask);
                    }
                }
            }
            u32x4 b[CW][4];
#pragma unroll
            for (int h = 0; h < CW; ++h)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (BITS == 8) b[h][t] = dq[h][t].frag(qv[0][t], qv[1][t]);
                    else if constexpr (BITS == 3) {
                        const unsigned lo = kg < 2 ? qv[0][t] : qv[1][t], hi = kg < 2 ? qv[1][t] : qv[2][t];
                        b[h][t] = dq[h][t].frag((unsigned)((((unsigned long long)hi << 32) | lo) >> w3_sh) & 0xFFFFFFu);
                    } else if constexpr (BITS == 2) {
                        const unsigned wsel = (kg & 2) ? qv[1][t] : qv[0][t];
                        b[h][t] = dq[h][t].frag((kg & 1) ? (wsel >> 16) : (wsel & 0xFFFFu));
                    } else b[h][t] = dq[h][t].frag(qv[h][t]);
                }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (XREG) *(u32x4*)(wbase + stage * SB + (WD + rt) * 1024 + lane * 16) = xr[stage][rt];
                const u32x4 x4 = *(const u32x4*)(st + (WD + rt) * 1024 + a_slot);
                u32x4 o;                                          // x in the slot order of the fragments: k0,k4,k1,k5,k2,k6,k3,k7
                o[0] = __builtin_amdgcn_perm(x4[2], x4[0], 0x05040100u);
                o[1] = __builtin_amdgcn_perm(x4[2], x4[0], 0x07060302u);
                o[2] = __builtin_amdgcn_perm(x4[3], x4[1], 0x05040100u);
                o[3] = __builtin_amdgcn_perm(x4[3], x4[1], 0x07060302u);
#pragma unroll
                for (int h = 0; h < CW; ++h)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[rt][h][t] = Mma16<T>::run(o, b[h][t], acc[rt][h][t]);
            }
        };
        auto wait_newer = [&](int newer) __attribute__((always_inline)) {     // stages issued after the one about to be consumed (wave-uniform)
            if constexpr (D >= 3) {
                if (newer >= 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VOPS) : "memory"); return; }
            }
            if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VOPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        stamp();
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (ws + d < we) issue(ws + d, d);
        stamp();
        for (int s0 = ws; s0 < we; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int s = s0 + d;
                if (s < we) {
                    wait_newer(min(D - 1, we - 1 - s));
                    stamp();
                    consume(s, d);
                    if (s + D < we) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // WAR: this stage's ds_reads are done
                        issue(s + D, d);
                    }
                    stamp();
                }
            }
        }
    }
#ifdef GPTQ_MID_TL
    {   // dump before the landing areas are reused as slabs: [workgroup][wave][128] = count, then the stamps
        stamp();
        unsigned long long* const tl_out = *(unsigned long long* const*)((const char*)p.err - 8 + 16);
        if (tl_out && lane == 0) {
            unsigned long long* o = tl_out + ((size_t)blockIdx.x * 8 + wave) * 128;
            o[0] = (unsigned long long)tl_n | ((unsigned long long)(we - ws) << 32);
            o[1] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < tl_n && i < 126; ++i) o[2 + i] = tl_lds[i];
        }
    }
    // epilogue stamps go straight to the buffer: entries [100 ..) of the wave's row
    unsigned long long* const tl_epi = *(unsigned long long* const*)((const char*)p.err - 8 + 16);
    int epi_n = 0;
    auto estamp = [&]() __attribute__((always_inline)) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (tl_epi && lane == 0 && epi_n < 20) tl_epi[((size_t)blockIdx.x * 8 + wave) * 128 + 100 + epi_n] = t;
        ++epi_n;
    };
#else
    auto estamp = [&]() __attribute__((always_inline)) {};
#endif

    // ---- cross-wave sum (LDS slabs over the landing areas, fixed order), CH row tiles at a time, then write / publish / combine ----------
    // C/D layout of the 16x16 MFMA: column = lane & 15 (-> strip column 4 j + t), row = 4 * (lane >> 4) + r.  A lane writes, per
    // (row tile, r), the float4 over t = its 4 adjacent columns of one row: lane-linear 16-byte LDS accesses both ways.
    constexpr int CH = RT < 4 ? RT : 4;
    constexpr int E = CH * 4 * 64;                                // float4 entries per chunk and wave
    f32x4* const slab = (f32x4*)smem;                             // [W][CH * 4][64]
    const size_t pslab = (size_t)p.M * p.nsum;
    bool partials_ready = (p.ksplit == 1 || ks != 0 || p.gran);
    unsigned tag = 0, ep = 0;
    if (p.ksplit > 1) {                                           // the tile's launch epoch: granule tags (tag = epoch + 1 in a NaN pattern) and flag stamps are derived from it
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ep) : "s"(p.epochs + tile) : "memory");
        tag = 0x7FE00000u | ((ep + 1u) & 0x1FFFFFu);
    }
    // Flag combine: a slice's flag word carries THIS launch's stamp (epoch + 1, never 0), not a bare 1.  A producer that arrives after its owner gave up
    // (bounded wait: another kernel or process held its CU) then leaves a stamp no later launch waits for, instead of a 1 that the next launch on this
    // tile would take for "published" before the partials are written (round-3 review: persistent silent corruption after one timeout).
    const unsigned fstamp = (ep + 1u) | 0x80000000u;
    bool gave_up = false;
    __syncthreads();                                              // every wave is done with its landing area
    estamp();                                                     // E0: first barrier passed
#pragma unroll
    for (int cc = 0; cc < ((RT + CH - 1) / CH) * CW; ++cc) {      // chunks: CH row tiles of one 64-column half
        const int c0 = (cc / CW) * CH, hh = cc % CW;
        if (cc) __syncthreads();                                  // the previous chunk's slabs have been read
#pragma unroll
        for (int rt = 0; rt < CH; ++rt) {
            if (c0 + rt < RT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[(size_t)wave * E + (rt * 4 + r) * 64 + lane] =
                        f32x4{acc[(c0 + rt) % RT][hh][0][r], acc[(c0 + rt) % RT][hh][1][r], acc[(c0 + rt) % RT][hh][2][r], acc[(c0 + rt) % RT][hh][3][r]};
            }
        }
        __syncthreads();
        if (!partials_ready) {                                    // owner slice, first chunk: every other slice of this strip has published
            if (tid < p.ksplit - 1) {
                const unsigned* f = p.flags + (size_t)tile * 8 + 1 + tid;
                unsigned v = 0;
                for (unsigned spins = 0;; ++spins) {
                    v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == fstamp) break;
                    if (spins > p.max_spins) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            partials_ready = true;
            estamp();                                             // (owner) E: flags seen
        }
        for (int e = tid; e < E; e += blockDim.x) {
            const int ln = e & 63, rr = e >> 6;                   // rr = rt * 4 + r inside the chunk
            const int rt = c0 + (rr >> 2);
            if (rt >= RT) continue;
            f32x4 v = slab[e];
            for (int w = 1; w < W; ++w) v += slab[(size_t)w * E + e];
            const int m = m0 + rt * 16 + 4 * (ln >> 4) + (rr & 3);
            const int n = strip * SC + hh * 64 + (ln & 15) * 4;
            if (m >= p.M) continue;
            if (p.ksplit > 1 && p.gran) {
                const size_t at = (size_t)m * p.nsum + sg.col0 + n;
                unsigned long long* const gbase = (unsigned long long*)p.partial;
                if (ks != 0) {
                    unsigned long long* g = gbase + (size_t)(ks - 1) * pslab + at;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        __hip_atomic_store(g + t, (unsigned long long)as_u32(v[t]) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
                }
                for (int k = 0; k + 1 < p.ksplit; ++k) {          // fixed order: bit-reproducible; each granule is taken the moment its tag is this launch's
                    const unsigned long long* g = gbase + (size_t)k * pslab + at;
                    unsigned long long a[4];
                    for (unsigned spins = 0;; ++spins) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) a[t] = __hip_atomic_load(g + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const bool ok = (unsigned)(a[0] >> 32) == tag && (unsigned)(a[1] >> 32) == tag && (unsigned)(a[2] >> 32) == tag && (unsigned)(a[3] >> 32) == tag;
                        if (ok) break;
                        if (spins > p.max_spins) { gave_up = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] += as_f32((unsigned)a[t]);
                    // consumed granules are cleared: between launches the exchange area holds NO valid tag, so a tag that is valid now was written
                    // by this launch -- per-strip epochs alone would let a strip whose epoch lags (it is used by fewer layers) accept what another
                    // layer published at the same address under the same number
                    unsigned long long* gw = gbase + (size_t)k * pslab + at;
#pragma unroll
                    for (int t = 0; t < 4; ++t) __hip_atomic_store(gw + t, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (p.ksplit > 1) {
                const size_t at = (size_t)m * p.nsum + sg.col0 + n;
                if (ks != 0) {
                    store16_sc1(p.partial + (size_t)(ks - 1) * pslab + at, v);
                    continue;
                }
                for (int k = 0; k + 1 < p.ksplit; ++k) {          // fixed order; sc1 loads bypass this XCD's non-coherent L2 lines
                    const unsigned long long* src = (const unsigned long long*)(p.partial + (size_t)k * pslab + at);
                    const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[0] += as_f32((unsigned)lo); v[1] += as_f32((unsigned)(lo >> 32));
                    v[2] += as_f32((unsigned)hi); v[3] += as_f32((unsigned)(hi >> 32));
                }
            }
            if (sg.bias) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] += DType<T>::to_f32(((const T*)sg.bias)[n + t]);
            }
            *(u32x2*)((T*)sg.out + (size_t)m * N + n) = pack4<T>(v);
        }
    }
    estamp();                                                     // E: all chunks reduced and written
    if (p.ksplit > 1 && p.gran) {
        if (gave_up) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ks == 0) {
            __syncthreads();                                      // every granule of this strip has been taken (so every producer has read the epoch)
            if (tid == 0) __hip_atomic_store(p.epochs + tile, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (p.ksplit > 1) {
        if (ks != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every publishing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.flags + (size_t)tile * 8 + ks, fstamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();                                      // every partial has been read (and every producer has read the epoch: its flag is up)
            if (tid < p.ksplit - 1) __hip_atomic_store(p.flags + (size_t)tile * 8 + 1 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) __hip_atomic_store(p.epochs + tile, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    estamp();                                                     // E: end
}

}  // namespace midk

// ---- host side ---------------------------------------------------------------------------------------------------------
static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// 128-column strips by default?  (filled in from tools/mid_sweep.py; s128 = strips of 128 columns over all layers of the launch)
static bool mid_prefers_wide_strips(int M, int s128) {
    (void)M; (void)s128;
    return false;
}

// Row blocks by default (filled in from tools/mid_sweep.py): 1 = the whole M in every workgroup
// Row blocks: two row tiles per workgroup, the blocks of a strip on adjacent workgroup ids (same XCD, same moment: the strip's weights are fetched
// from HBM once and shared through the L2).  Measured (profiles/r03_mid_kernel_row_blocks_adjacent.log; us, one block -> row blocks): 4096x4096 M = 64
// 14.4 -> 11.5 (2 blocks x 2 K slices), M = 128 23.0 -> 13.7 (4 blocks, no K split); 11008x4096 M = 64 21.7 -> 20.3, M = 128 34.4 -> 29.1.  Only when
// the blocks fill the chip -- 192..256 workgroups with at most 2 K slices: 5120x5120 with 2 blocks is 160 workgroups (17.2 = no gain) and with 4 is 320
// (24.7 against 23.0 for the tiled kernel).
static int mid_row_blocks(int M, int strips, int K) {
    const int t16 = (M + 15) / 16;
    if (strips >= 160 || t16 < 3) return 1;
    if (t16 > 8) {                                     // 129..256 rows: blocks of four row tiles, short K only (profiles/r03_mid_kernel_row_blocks_192_256.log:
        if (K > 5120) return 1;                        // 4096x4096 M = 192 / 256 22.6 / 22.9 (tiled) -> 18.5 / 18.9, 5120x5120 M = 192 28.8 -> 23.9; long K: no gain)
        const int b4 = (t16 + 3) / 4, t4 = strips * b4;
        return (t4 >= 192 && t4 <= 256) ? b4 : 1;
    }
    const int blocks = (t16 + 1) / 2;
    const int tiles = strips * blocks;
    const int ks = tiles >= 160 ? 1 : 256 / tiles;
    const int wgs = tiles * ks;
    if (wgs < 224 && K > 5120) return 1;               // 11008x4096 M = 96: 3 blocks = 192 workgroups 29.2 against 28.1 with one block x 4 slices
    return (wgs >= 192 && wgs <= 256 && ks <= 2) ? blocks : 1;
}

// 8-bit layers by default?  (filled in from tools/nonq4_batched.py)
// Measured (tools/nonq4_batched.py, profiles/r03_nonq4_batched_int8.log; us, previous default -> this kernel, int8 g32):
//   4096x4096   M = 8 / 16 / 64 / 128: 12.3 / 16.3 / 17.8 / 26.5 -> 11.1 / 11.3 / 13.5 / 17.3
//   4096x11008  M = 8 / 16 / 64 / 128: 23.7 / 24.2 / 29.3 / 39.5 -> 15.1 / 16.3 / 26.6 / 41.4 (tiled stays at 128)
//   11008x4096  M = 8 / 16 / 64 / 128: 27.1 / 32.8 / 32.4 / 44.1 -> 16.1 / 16.7 / 26.4 / 38.0
static bool mid_pays_int8(int M, int strips, int K) {
    (void)K;
    return M >= 5 && M <= 128 && !(M > 64 && strips >= 160);
}

// 2-bit layers by default?  (filled in from tools/nonq4_batched.py)
// Measured (tools/nonq4_batched.py, profiles/r03_nonq4_batched_int2.log; us, previous default -> this kernel, int2 g64, M = 8 / 16 / 64 / 128):
//   4096x4096 10.2 / 11.3 / 15.4 / 33.2 -> 9.0 / 9.0 / 12.1 / 15.8;  4096x11008 24.9 / 26.2 / 33.9 / 43.3 -> 13.3 / 13.1 / 20.3 / 34.0;
//   11008x4096 23.5 / 23.5 / 28.7 / 50.0 -> 13.6 / 13.9 / 21.6 / 32.7
static bool mid_pays_int2(int M, int strips, int K) {
    (void)strips; (void)K;
    return M >= 5 && M <= 128;
}

// 3-bit layers by default?  (filled in from tools/nonq4_batched.py)
// Measured (tools/nonq4_batched.py, profiles/r03_nonq4_batched_int3.log; us, previous default -> this kernel, int3 g32):
//   4096x4096   M = 8 / 16 / 64 / 128:  9.0 / 10.8 / 15.1 / 29.9 -> 10.4 (GEMV stays) / 10.3 / 13.0 / 17.6
//   4096x11008  M = 8 / 16 / 64 / 128: 17.7 / 26.9 / 32.9 / 38.6 -> 15.0 / 15.3 / 21.8 / 36.5
//   11008x4096  M = 8 / 16 / 64 / 128: 16.7 / 21.6 / 30.4 / 44.7 -> 15.1 (GEMV stays: equal) / 15.7 / 24.2 / 38.7
static bool mid_pays_int3(int M, int strips, int K) {
    (void)K;
    return M <= 128 && (M >= 9 || (M >= 5 && strips >= 160));
}

MidPlan plan_mid(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* tune) {
    MidPlan pl{};
    if (n < 1 || n > 4 || M < 1 || M > 256) return pl;
    const gptq_layer_t& A = *Ls[0];
    int strips = 0, nsum = 0;
    const int rt16 = (M + 15) / 16;
    const int rt_ = rt16 <= 2 ? 2 : (rt16 <= 4 ? 4 : (rt16 <= 6 ? 6 : 8));   // (whole M in one block: what the 128-column-strip experiment is defined for)
    bool all128 = true;
    for (int i = 0; i < n; ++i) all128 = all128 && Ls[i]->N % 128 == 0;
    // 128-column strips (two 64-column halves per wave): forced by tuning.reserved[3] = 2 (1 forces 64), else by measurement (below)
    const int want_cw = (tune && tune->reserved[3] > 0) ? tune->reserved[3] : 0;
    int cw = (want_cw == 2 && rt_ <= 4 && all128 && A.bits == 4) ? 2 : 1;
    if (want_cw == 0 && rt_ <= 4 && all128 && A.bits == 4) {
        int s128 = 0;
        for (int i = 0; i < n; ++i) s128 += Ls[i]->N / 128;
        if (mid_prefers_wide_strips(M, s128)) cw = 2;
    }
    for (int i = 0; i < n; ++i) {
        const gptq_layer_t& L = *Ls[i];
        if ((L.bits != 4 && L.bits != 8 && L.bits != 3 && L.bits != 2) || L.bits != A.bits || (L.bits == 3 && L.N % 128) || (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16) || L.epilogue != GPTQ_EPI_NONE) return pl;
        if (L.K % 32 || L.N % 64 || L.group_size % 32) return pl;
        if (L.g_idx != nullptr && (n > 1 || !L.qweight_seq || !L.perm)) return pl;      // act-order: single layers only (x is permuted per layer)
        if (L.K != A.K || L.group_size != A.group_size || L.dtype != A.dtype || L.zero_mode != A.zero_mode) return pl;
        strips += L.N / (64 * cw);
        nsum += L.N;
    }
    const int lg = ilog2_exact(A.group_size / 32);
    if (lg < 0) return pl;
    if ((size_t)M * A.K * 2 >= ((size_t)1 << 31)) return pl;                            // 32-bit lane offsets into x
    pl.nseg = n;
    pl.strips_total = strips;
    pl.nsum = nsum;
    // row blocks: tuning.lanes_n (unused by this kernel otherwise) forces the count; default by measurement (mid_row_blocks)
    int rbs = (tune && tune->lanes_n > 0 && tune->path == 3) ? tune->lanes_n : mid_row_blocks(M, strips, A.K);
    if (rbs < 1) rbs = 1;
    if (rbs > rt16) rbs = rt16;
    {
        const int per = (rt16 + rbs - 1) / rbs;                                         // row tiles per block
        if (per > 8) return pl;                                                         // 129+ rows need row blocks
        const int rtb = per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : (per <= 6 ? 6 : 8)));
        if (cw == 2 && rtb > 4) return pl;
        pl.rt = rtb;
        pl.row_blocks = (rt16 + rtb - 1) / rtb;                                         // no empty block
    }
    pl.cw = cw;
    pl.lg_gsteps = lg;
    const int S = A.K / 32;
    pl.ksteps_total = S;
    pl.waves = 8;
    int ks = (tune && tune->ksplit > 0 && tune->path == 3) ? tune->ksplit : 0;
    const int tiles = strips * pl.row_blocks;
    if (!ks) {
        ks = tiles >= 160 ? 1 : 256 / tiles;                                            // one workgroup per CU is one round
        if (ks > 8) ks = 8;
        while (ks > 1 && S / ks < 2 * pl.waves) --ks;                                   // at least two K-steps per wave
    }
    if (ks > 8) ks = 8;
    while (ks > 1 && (long)tiles * ks > 256) --ks;                                     // the owner slice WAITS for the others: every workgroup of the launch must be resident (one per CU, 256 CUs)
    if (ks > S) ks = S;
    if (ks > 1 && (size_t)tiles * 8 * 4 > WS_HEADER_EPOCH_OFFSET) ks = 1;             // 8 flag words per (row block, strip) in the ticket half of the header
    if (ks > 1 && (size_t)tiles * 4 > WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES - WS_HEADER_EPOCH_OFFSET) ks = 1;   // ... and one epoch word per tile in its second half (flag stamps)
    if (ks < 1) ks = 1;
    pl.ksteps_per_split = (S + ks - 1) / ks;
    pl.ksplit = (S + pl.ksteps_per_split - 1) / pl.ksteps_per_split;                    // no empty slices
    int stages = (tune && tune->reserved[0] > 0) ? tune->reserved[0] : ((pl.rt == 2 && strips >= 160) ? 3 : 2);   // tools/mid_sweep.py: two, except two row tiles on wide layers (4096 x 11008, M = 17: 12.3 against 12.8 us)
    if (stages > 3) stages = 3;
    if (stages < 2 || A.bits != 4) stages = 2;
    const int ch = pl.rt < 4 ? pl.rt : 4;
    // LDS: waves x (stages x (1 + rt) KiB + group table); the table grows with a wave's K range, so when 8 waves do not fit first drop the third
    // stage, then waves (8 row tiles with one long K slice: 7 waves)
    for (;;) {
        const int spw = (pl.ksteps_per_split + pl.waves - 1) / pl.waves;                // K-steps per wave
        const int groups = ((spw - 1) >> lg) + 2;                                       // groups a wave's range can touch (unaligned start)
        pl.tab_bytes = ((groups * cw + 3) / 4) * 1024;                                  // 256 B per (group, 64-column half); the table DMA writes whole KiB
#ifdef GPTQ_MID_TL
        pl.tab_bytes += 1024;                                                           // lab build: the wave's stamp area
#endif
        const size_t land = (size_t)pl.waves * ((size_t)stages * ((A.bits == 8 ? 2 : cw) + pl.rt) * 1024 + pl.tab_bytes);   // 3-bit: one (768-byte) weight slot as 4-bit
        const size_t slabs = (size_t)pl.waves * ch * 4096;
        pl.lds_bytes = (land > slabs ? land : slabs) + 16;
        if (pl.lds_bytes <= 160 * 1024) break;
        if (stages > 2) --stages;
        else if (pl.waves > 4) --pl.waves;
        else break;
    }
    pl.stages = stages;
    pl.xreg = tune && tune->reserved[1] == 1 && A.bits == 4;
    pl.bits = A.bits;
    if (pl.lds_bytes > 160 * 1024) return pl;
    // experiment knob: reserved[1] = 3 -> the granule combine.  One hop instead of three, but 8-byte write-through stores and polls for every VALUE of a
    // 16..128 x 64 tile: 4096^2 M = 64 27.7 us against 14.8 with flags, M = 128 46 against 23 (profiles/r03_mid_kernel_granules_vs_flags.log).  What pays
    // for the one to four rows of the streamed GEMV (64 values per strip) does not scale to a tile.
    pl.gran = tune && tune->reserved[1] == 3 &&
              (size_t)strips * pl.row_blocks * 4 <= WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES - WS_HEADER_EPOCH_OFFSET;   // one epoch word per (row block, strip) in the header's second half
    pl.partial_bytes = pl.ksplit > 1 ? (size_t)(pl.ksplit - 1) * M * nsum * (pl.gran ? 8 : 4) : 0;
#ifdef GPTQ_MID_TL
    if (!pl.partial_bytes) pl.partial_bytes = 256;                                      // lab build: always a workspace (its header tail carries the timeline pointer)
#endif
    // Measured preference (tools/mid_sweep.py, us per launch, planner's previous choice -> this kernel; M = 17 / 33 / 64 / 96 / 128):
    //   4096x4096   11.7 / 13.5 / 15.7 / 22.5 / 24.7 ->  9.8 / 12.2 / 14.0 / 18.4 / 22.2
    //   4096x11008  14.5 / 19.3 / 22.6 / 26.5 / 27.9 -> 12.3 / 16.5 / 21.1 / 26.7 / 27.1      (172 strips: the tiled kernel keeps 65+ rows)
    //   11008x4096  15.7 / 20.4 / 24.5 / 33.8 / 35.6 -> 13.4 / 17.5 / 21.3 / 27.5 / 33.3
    // Up to 16 rows gemm_stream64_kernel's one-row-tile form (16 waves) stays.
    int nmax = 0;
    for (int i = 0; i < n; ++i) nmax = Ls[i]->N > nmax ? Ls[i]->N : nmax;
    // Other shapes (profiles/r03_mid_kernel_more_shapes.log): 17..64 rows win on 5120^2, 8192^2, 3584x8192, 8192x3584, 13824x5120, 28672x8192 (5-20 %);
    // 97..128 rows only on the 64-strip layers above (5120^2: 26.4 against 23.1, 8192x3584: 28.8 against 25.6); layers of < 32 strips keep
    // the skinny kernel from 33 rows (8192x1024 M = 64: 15.9 against 14.0); very wide layers keep the tiled kernel from 33 rows.
    if (A.bits != 4) {                                 // 3- / 8-bit (tools/nonq4_batched.py): measured preference, see plan_gemm
        pl.pays = A.bits == 8 ? mid_pays_int8(M, strips, A.K) : (A.bits == 3 ? mid_pays_int3(M, strips, A.K) : mid_pays_int2(M, strips, A.K));
        pl.ok = true;
        return pl;
    }
    if (n >= 2) {
        // several layers sharing x (tools/mid_multi_ab.py, profiles/r03_mid_kernel_multi_layer_ab.log; us, one launch against layer by layer): q|k|v
        // M = 33 / 64 / 96 / 128: 17.3 / 22.1 / 28.1 / 35.9 against 34.4 / 35.7 / 40.3 / 42.5; gate|up 32.3 / 41.3 / 51.7 against 34.7 / 43.6 / 53.1, and
        // 58-68 against 55.9 at 128 rows (its layers are 172 strips each: the tiled kernel's regime)
        int smax = 0;
        for (int i = 0; i < n; ++i) smax = Ls[i]->N / (64 * cw) > smax ? Ls[i]->N / (64 * cw) : smax;
        pl.pays = M >= 17 && M <= 128 && (M <= 96 || smax < 160) && !(M > 32 && nmax >= 12288);
        pl.ok = true;
        return pl;
    }
    // (late round 6, tools/mid_band_sweep.py: beyond 128 Mi weights the 128 x 256-tile kernel wins above 64 rows -- 28672x8192 at 96 rows 85.2 against 69.3 us)
    pl.pays = M >= 17 && (M <= 128 || pl.row_blocks > 1) && !(M > 64 && strips >= 160) && !(M > 32 && nmax >= 12288) && !(M > 32 && strips < 32) &&
              !(M > 96 && strips != 64 && pl.row_blocks == 1) && !(M > 64 && (size_t)A.K * nmax > ((size_t)128 << 20));
    pl.ok = true;
    return pl;
}

template <typename T, int RT, int D>
static hipError_t launch_mid_one(const MidPlan& pl, const midk::MidParams& p, hipStream_t st) {
    const dim3 grid(pl.strips_total * pl.row_blocks * pl.ksplit), block(pl.waves * 64);
    if (pl.bits != 4) {
        if constexpr (D == 2) {
            if (pl.bits == 8) hipLaunchKernelGGL((midk::gemm_mid_kernel<T, RT, 2, false, 1, 8>), grid, block, pl.lds_bytes, st, p);
            else if (pl.bits == 3) hipLaunchKernelGGL((midk::gemm_mid_kernel<T, RT, 2, false, 1, 3>), grid, block, pl.lds_bytes, st, p);
            else hipLaunchKernelGGL((midk::gemm_mid_kernel<T, RT, 2, false, 1, 2>), grid, block, pl.lds_bytes, st, p);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if constexpr (RT <= 4) {
        if (pl.cw == 2) {
            hipLaunchKernelGGL((midk::gemm_mid_kernel<T, RT, D, false, 2>), grid, block, pl.lds_bytes, st, p);
            return hipGetLastError();
        }
    }
    if (pl.xreg) hipLaunchKernelGGL((midk::gemm_mid_kernel<T, RT, D, true>), grid, block, pl.lds_bytes, st, p);
    else hipLaunchKernelGGL((midk::gemm_mid_kernel<T, RT, D, false>), grid, block, pl.lds_bytes, st, p);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_mid_t(const MidPlan& pl, const midk::MidParams& p, hipStream_t st) {
    switch (pl.rt * 4 + pl.stages) {
        case 1 * 4 + 2: return launch_mid_one<T, 1, 2>(pl, p, st);
        case 1 * 4 + 3: return launch_mid_one<T, 1, 3>(pl, p, st);
        case 2 * 4 + 2: return launch_mid_one<T, 2, 2>(pl, p, st);
        case 2 * 4 + 3: return launch_mid_one<T, 2, 3>(pl, p, st);
        case 4 * 4 + 2: return launch_mid_one<T, 4, 2>(pl, p, st);
        case 4 * 4 + 3: return launch_mid_one<T, 4, 3>(pl, p, st);
        case 6 * 4 + 2: return launch_mid_one<T, 6, 2>(pl, p, st);
        case 8 * 4 + 2: return launch_mid_one<T, 8, 2>(pl, p, st);
        default: return hipErrorInvalidValue;
    }
}

namespace mlpk { extern int g_cu_count[64]; }     // per device ordinal, filled by gptq_init (mlp.hip)

hipError_t launch_mid(const gptq_layer_t* const* Ls, const MidPlan& pl, const void* x, void* const* outs, int M, void* ws_header, void* partial,
                      const uint32_t* qweight_override, hipStream_t st) {
    if (!pl.ok) return hipErrorNotSupported;
    if (pl.ksplit > 1 && (!ws_header || !partial)) return hipErrorInvalidValue;
    if (pl.ksplit > 1) {
        // the owner slice of a strip WAITS (bounded) for the other slices: every workgroup of the launch must be resident at once -- one per CU (its
        // LDS allows no second).  The planner assumes MI355X's 256 CUs; a device with fewer (partitioned, masked) is refused here, loudly.
        // (an unknown CU count -- gptq_init() not run on this device -- is refused too: the check must not be skipped silently)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || mlpk::g_cu_count[dev] <= 0) return hipErrorNotInitialized;
        if (pl.strips_total * pl.row_blocks * pl.ksplit > mlpk::g_cu_count[dev]) return hipErrorLaunchOutOfResources;
    }
    midk::MidParams p{};
    int blk = 0, col = 0;
    for (int i = 0; i < pl.nseg; ++i) {
        const gptq_layer_t& L = *Ls[i];
        midk::MidSeg& sg = p.seg[i];
        sg.qweight = (i == 0 && qweight_override) ? qweight_override : L.qweight;
        sg.qzeros = L.qzeros;
        sg.scales = L.scales;
        sg.bias = L.bias;
        sg.out = outs[i];
        sg.N = L.N;
        blk += L.N / (64 * pl.cw);
        sg.blk_end = blk;
        sg.col0 = col;
        col += L.N;
    }
    p.x = x;
    p.partial = (float*)partial;
    p.flags = (unsigned*)ws_header;
    p.epochs = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_EPOCH_OFFSET) : nullptr;
    p.gran = pl.gran ? 1 : 0;
    p.err = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2 : nullptr;
    p.nseg = pl.nseg; p.M = M; p.K = Ls[0]->K; p.zero_mode = Ls[0]->zero_mode;
    p.ksplit = pl.ksplit; p.ksteps_per_split = pl.ksteps_per_split; p.nsum = pl.nsum;
    p.lg_gsteps = pl.lg_gsteps; p.tab_bytes = pl.tab_bytes; p.strips_total = pl.strips_total; p.row_blocks = pl.row_blocks;
    p.max_spins = 1u << 22;
    return (Ls[0]->dtype == GPTQ_F16) ? launch_mid_t<f16>(pl, p, st) : launch_mid_t<bf16>(pl, p, st);
}

template <typename T, int RT, int D> static hipError_t grant_mid() {
    hipError_t e = hipFuncSetAttribute((const void*)midk::gemm_mid_kernel<T, RT, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)midk::gemm_mid_kernel<T, RT, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (RT <= 4) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)midk::gemm_mid_kernel<T, RT, D, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if constexpr (D == 2) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)midk::gemm_mid_kernel<T, RT, 2, false, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)midk::gemm_mid_kernel<T, RT, 2, false, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)midk::gemm_mid_kernel<T, RT, 2, false, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    return e;
}
template <typename T> static hipError_t grant_mid_t() {
    hipError_t e = hipSuccess;
    auto acc = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    acc(grant_mid<T, 1, 2>()); acc(grant_mid<T, 1, 3>()); acc(grant_mid<T, 2, 2>()); acc(grant_mid<T, 2, 3>()); acc(grant_mid<T, 4, 2>()); acc(grant_mid<T, 4, 3>());
    acc(grant_mid<T, 6, 2>()); acc(grant_mid<T, 8, 2>());
    return e;
}
hipError_t init_gemm_mid_device() {
    hipError_t e = grant_mid_t<f16>();
    if (e == hipSuccess) e = grant_mid_t<bf16>();
    return e;
}

}  // namespace gptq
