/* CPU oracle (plain C restatement) for the GPTQ QuantLinear hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Never linked into, loaded by, or called from the product (autogptq_amd / libgptq_mi355x.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Scalar, one-value-at-a-time statement of the reference's integer unpack and dequant, written
 * to be obviously right rather than fast, and structured differently from oracle/gptq_oracle.py
 * (bit-stream addressing instead of per-word shift tables) so the two check each other.
 * Reference being restated (AutoGPTQ v0.8.0.dev0):
 *   layout / pack      auto_gptq/nn_modules/qlinear/qlinear_cuda.py:135-203
 *   unpack "wrap"      auto_gptq/nn_modules/qlinear/qlinear_cuda_old.py:295-316 (3-bit :317-344)
 *   unpack "nowrap"    auto_gptq/nn_modules/qlinear/qlinear_cuda.py:257-295
 *   dequant + matmul   qlinear_cuda_old.py:348-350, qlinear_cuda.py:302,313
 *
 * A packed column (qweight) or packed row (qzeros) is a little-endian bit stream: value v
 * occupies bits [bits*v, bits*v + bits) of the stream formed by concatenating the 32-bit words
 * in order.  For 2/4/8-bit no value crosses a word; for 3-bit, values 10 and 21 of every 32 do
 * (qlinear_cuda.py:144-162) -- which is exactly what the stream view gives.
 */
#include <stdint.h>
#include <stddef.h>

static inline uint32_t stream_field(const uint32_t *words, size_t stride, size_t v, int bits)
{
    size_t bit = (size_t)bits * v;
    size_t w = bit >> 5;
    unsigned sh = (unsigned)(bit & 31);
    uint64_t lo = words[w * stride];
    uint64_t hi = (sh + (unsigned)bits > 32) ? words[(w + 1) * stride] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> sh) & ((1u << bits) - 1u));
}

/* w[k,n] in [0, maxq]; qweight is int32 [K/32*bits, N] row-major */
int gptq_oracle_unpack_weights(const int32_t *qweight, int K, int N, int bits, uint16_t *w_out)
{
    if (!(bits == 2 || bits == 3 || bits == 4 || bits == 8) || K % 32) return 1;
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n)
            w_out[(size_t)k * N + n] =
                (uint16_t)stream_field((const uint32_t *)qweight + n, (size_t)N, (size_t)k, bits);
    return 0;
}

/* zero_mode 0 = wrap: (f+1)&maxq ; 1 = nowrap: f+1.  qzeros is int32 [G, N/32*bits] */
int gptq_oracle_unpack_zeros(const int32_t *qzeros, int G, int N, int bits, int zero_mode, int32_t *z_out)
{
    if (!(bits == 2 || bits == 3 || bits == 4 || bits == 8) || N % 32) return 1;
    size_t row_words = (size_t)N / 32 * bits;
    for (int g = 0; g < G; ++g)
        for (int n = 0; n < N; ++n) {
            int32_t f = (int32_t)stream_field((const uint32_t *)qzeros + g * row_words, 1, (size_t)n, bits);
            int32_t z = f + 1;
            if (zero_mode == 0) z &= (1 << bits) - 1;
            z_out[(size_t)g * N + n] = z;
        }
    return 0;
}

/* y[m,n] = sum_k x[m,k] * scales[g(k),n] * (w[k,n] - z[g(k),n]) (+bias), double accumulation,
 * exact (unrounded) dequant.  scales/x/bias given as float; g_idx may be NULL (k / group_size). */
int gptq_oracle_forward_f64(const float *x, const int32_t *qweight, const int32_t *qzeros,
                            const float *scales, const int32_t *g_idx, const float *bias,
                            int M, int K, int N, int bits, int group_size, int zero_mode, double *y)
{
    if (!(bits == 2 || bits == 3 || bits == 4 || bits == 8) || K % 32 || N % 32) return 1;
    size_t row_words = (size_t)N / 32 * bits;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) y[(size_t)m * N + n] = bias ? (double)bias[n] : 0.0;
    for (int k = 0; k < K; ++k) {
        int g = g_idx ? g_idx[k] : k / group_size;
        for (int n = 0; n < N; ++n) {
            int32_t w = (int32_t)stream_field((const uint32_t *)qweight + n, (size_t)N, (size_t)k, bits);
            int32_t z = (int32_t)stream_field((const uint32_t *)qzeros + g * row_words, 1, (size_t)n, bits) + 1;
            if (zero_mode == 0) z &= (1 << bits) - 1;
            double dq = (double)scales[(size_t)g * N + n] * (double)(w - z);
            for (int m = 0; m < M; ++m) y[(size_t)m * N + n] += (double)x[(size_t)m * K + k] * dq;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The decode copy (include/gptq_mi355x.h, gptq_prepack_decode): the library's own load-time
 * re-layout, the role exllamav2's shuffle (exllamav2/cuda/q_matrix.cu:19-42) and Marlin's
 * repack (marlin/marlin_repack.cu:8-92) play in the reference.  Stated here DESTINATION-first
 * (for every stored bit: which source value and which of its bits), where gptq_oracle.py
 * states it source-first with reshapes -- the two check each other, and both check the
 * device kernels.
 *
 * A lane owns KPL consecutive k (32; 16 at 8 bits) of one column and stores them in WPL words
 * (4; 3 at 3 bits; 2 at 2 bits).  lane_value_of_bit answers: stored bit b of word j of a lane is bit *vb of
 * the lane's value number *vk.
 * ------------------------------------------------------------------------------------------ */
static void lane_value_of_bit(int bits, int j, int b, int *vk, int *vb)
{
    if (bits == 4) {                     /* nibble p of word j holds k = 8 j + (0 2 4 6 1 3 5 7)[p] */
        int p = b >> 2;
        *vk = 8 * j + (p < 4 ? 2 * p : 2 * (p - 4) + 1);
        *vb = b & 3;
    } else if (bits == 8) {              /* byte p of word j holds k = 4 j + (0 2 1 3)[p] */
        int p = b >> 3;
        *vk = 4 * j + (p == 1 ? 2 : p == 2 ? 1 : p);
        *vb = b & 7;
    } else if (bits == 2) {              /* word j: bit pair 2 p of the low / high half = k 16 j + 2 p / + 2 p + 1 */
        int half = b >> 4, r = b & 15;
        *vk = 16 * j + 2 * (r >> 1) + half;
        *vb = r & 1;
    } else {                             /* 3 bits: halves hold the even / odd k of pairs 5 j .. 5 j + 4; bit 15 / 31 = bit j of k 30 / 31 */
        int half = b >> 4, r = b & 15;
        if (r == 15) { *vk = 30 + half; *vb = j; }
        else         { *vk = 2 * (5 * j + r / 3) + half; *vb = r % 3; }
    }
}

/* out: uint32 [N/16][chunks][4][16][WPL], chunks = ceil(K / (4 KPL)); k >= K read as 0 */
int gptq_oracle_decode_copy_weights(const int32_t *qweight, int K, int N, int bits, uint32_t *out)
{
    if (!(bits == 2 || bits == 3 || bits == 4 || bits == 8) || K % 32 || N % 16) return 1;
    const int kpl = bits == 8 ? 16 : 32, wpl = bits == 3 ? 3 : (bits == 2 ? 2 : 4);
    const int chunks = (K + 4 * kpl - 1) / (4 * kpl);
    size_t o = 0;
    for (int s = 0; s < N / 16; ++s)
        for (int c = 0; c < chunks; ++c)
            for (int slot = 0; slot < 4; ++slot)
                for (int col = 0; col < 16; ++col)
                    for (int j = 0; j < wpl; ++j, ++o) {
                        uint32_t word = 0;
                        for (int b = 0; b < 32; ++b) {
                            int vk, vb;
                            lane_value_of_bit(bits, j, b, &vk, &vb);
                            long k = ((long)c * 4 + slot) * kpl + vk;
                            if (k >= K) continue;
                            uint32_t v = stream_field((const uint32_t *)qweight + 16 * s + col, (size_t)N, (size_t)k, bits);
                            word |= ((v >> vb) & 1u) << b;
                        }
                        out[o] = word;
                    }
    return 0;
}

/* out: bytes [N/16][G][REC]; REC = 48 (16 x u16 scale bits, 16 x u8 zero) or, at 8 bits, 64 (16 x u16 zero).
 * scale_bits = the checkpoint's 16-bit scales verbatim, [G, N]. */
int gptq_oracle_decode_copy_consts(const int32_t *qzeros, const uint16_t *scale_bits, int G, int N, int bits,
                                   int zero_mode, uint8_t *out)
{
    if (!(bits == 2 || bits == 3 || bits == 4 || bits == 8) || N % 32) return 1;
    const int rec = bits == 8 ? 64 : 48;
    const size_t row_words = (size_t)N / 32 * bits;
    for (int s = 0; s < N / 16; ++s)
        for (int g = 0; g < G; ++g) {
            uint8_t *r = out + ((size_t)s * G + g) * rec;
            for (int col = 0; col < 16; ++col) {
                int n = 16 * s + col;
                uint16_t sb = scale_bits[(size_t)g * N + n];
                int32_t z = (int32_t)stream_field((const uint32_t *)qzeros + g * row_words, 1, (size_t)n, bits) + 1;
                if (zero_mode == 0) z &= (1 << bits) - 1;
                r[2 * col] = (uint8_t)(sb & 255);
                r[2 * col + 1] = (uint8_t)(sb >> 8);
                if (bits == 8) {
                    r[32 + 2 * col] = (uint8_t)(z & 255);
                    r[32 + 2 * col + 1] = (uint8_t)(z >> 8);
                } else
                    r[32 + col] = (uint8_t)z;
            }
        }
    return 0;
}
