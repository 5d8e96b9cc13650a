// gemv_tiled.hip -- decode (M <= 4; 5..8 rows where it pays: capi.hip, want_tiled) from the load-time DECODE COPY of a 3-, 4- or 8-bit layer, plain or act-order (round 4).  The kernel itself:
// gemv_tiled_kernel.cuh (instantiated here for plain layers, in gemv_tiled_act.hip for act-order ones, in gemv_tiled_peer.hip with the tensor-parallel
// epilogue); this file: the planner and the launch.  The 4-bit layout as the example (the other packings: TiledFmt, gptq_mi355x.h):
//
// What the reference does at load time in every fast backend -- exllamav2 shuffle_kernel (autogptq_extension/exllamav2/cuda/q_matrix.cu:19-42,
// called from :149), exllama make_sequential (exllama/cuda_func/q4_matrix.cu:105-169), Marlin gptq_repack + its scale permutation
// (marlin/marlin_repack.cu:8-92, qlinear_marlin.py:133-176) -- is a re-layout of the packed weights for the kernel that streams them.  Here the
// re-layout goes into NON-PERSISTENT side buffers (gptq_prepack_decode, csrc/utils.hip); the checkpoint tensors stay intact:
//
//   qweight_tiled  [strip s = 16 output columns][chunk c = 16 packed rows = 128 k][k-slot kb = 0..3][column 0..15][word w = 0..3]
//                  word (c, kb, col, w) = nibble_shuffle( qweight[16 c + 4 kb + w][16 s + col] ),  stored nibbles = k0 k2 k4 k6 k1 k3 k5 k7
//                  -> a strip is ONE contiguous run; a chunk is one contiguous KiB = one wave load; LANE (kb, col) of that load holds 4 consecutive
//                  packed rows (32 k) of ONE column; the magic-number extraction (q & 0x000f000f etc.) yields the k pairs (k0,k1)(k2,k3)(k4,k5)(k6,k7),
//                  the order x lies in memory (no v_perm on the activations).  Rows past K/8 are zero words (whole chunks).
//   qconst_tiled   [strip][group g][48 bytes] = 16 scales (layer dtype) + 16 one-byte zero-points AS USED (zero_mode applied): the constants of a
//                  strip are one contiguous run (1.5 KiB for K = 4096, g128) instead of 32-byte / 8-byte pieces of row-major [G, N] arrays.
//
// tools/membench2.hip (profiles/r04_membench_stripmajor.log): reading the same bytes as contiguous strips instead of 64-byte row segments 16 KiB apart
// takes 2.74 instead of 3.40 us (4096^2), 4.93 / 6.75 (11008x4096), 5.13 / 6.82 (q|k|v), 8.09 / 11.14 (gate|up) -- the rate of a flat stream.
// tools/declab.hip (profiles/r04_declab_*.log) then compared lane decompositions on that stream: 4 columns x 1 row per lane (every row its own group:
// a constant set per 32 weights, x re-read from L2 per row) against this one, us per launch: 4.37 -> 4.13, 7.90 -> 6.84, 8.12 -> 6.95, 12.95 -> 10.70.
//
// Kernel (one strip, or one K slice of it, per workgroup; W waves x U chunks in flight per wave):
//   * x (M rows, the slice's k range) and the strip's constants go global -> LDS by DMA FIRST (they return first), the wave's U weight loads behind them;
//     one barrier, by which time the weights are in flight;
//   * per chunk and lane: 16 bytes of weights (registers, nontemporal), x fragments by 4 ds_read_b128 (lane i of a 4-lane group reads x row i: the A
//     operand of v_mfma_f32_4x4x4), scale + zero-point by ds_read_u16 / ds_read_u8; w - z exactly in packed fp16, 8 matrix-core steps into one fp32
//     group sum, times the scale: the arithmetic of every other decode kernel here (one-hot rows return the reference's exact scales * (w - z));
//   * a lane owns ONE column: k-slots by two shuffles, waves through LDS, K slices through {fp32, tag} granules (stream_finish, gemv_shared.cuh);
//   * up to four layers that share x in one launch (gptq_forward_multi).
#include "gemv_tiled_kernel.cuh"

namespace gptq {

hipError_t launch_tiled_act(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st);      // gemv_tiled_act.hip
hipError_t init_gemv_tiled_act_device();
hipError_t launch_tiled_peer(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st);     // gemv_tiled_peer.hip
hipError_t init_gemv_tiled_peer_device();
hipError_t launch_tiled_pair(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st);     // gemv_tiled_pair.hip
hipError_t init_gemv_tiled_pair_device();
hipError_t launch_tiled_multi(const TiledPlan& pl, const TiledParams& p, int dtype, hipStream_t st);    // gemv_tiled_multi.hip
hipError_t init_gemv_tiled_multi_device();

// ---- plan + launch ---------------------------------------------------------------------------------------------------------------------
int tiled_max_waves(int u, int nstr) { return (u == 2 || nstr == 4) ? 16 : 8; }       // = tiled_maxw<U, XM>() of gemv_tiled_kernel.cuh: the launch bound each depth is compiled for
static int tiled_kpl(int bits) { return bits == 8 ? 16 : 32; }          // k per lane and chunk
static int tiled_chunk_bytes(int bits) { return bits == 3 ? 768 : (bits == 2 ? 512 : 1024); }
static int tiled_rec_bytes(int bits) { return bits == 8 ? 64 : 48; }

// A [gate | up] layer with the SILU_MUL epilogue: a plain (no act-order) layer whose halves are whole strips
static bool tiled_pair_layer(const gptq_layer_t& L) { return L.epilogue == GPTQ_EPI_SILU_MUL && L.g_idx == nullptr && L.N % (2 * GPTQ_STRIP_COLS) == 0; }

bool tiled_layer_ok(const gptq_layer_t& L) {
    if (!L.qweight_tiled || !L.qconst_tiled || L.tiled_cols != GPTQ_STRIP_COLS || (L.epilogue != GPTQ_EPI_NONE && !tiled_pair_layer(L))) return false;
    if (L.bits != 4 && L.bits != 8 && L.bits != 3 && L.bits != 2) return false;
    if (L.bits == 2 && L.epilogue != GPTQ_EPI_NONE) return false;                 // 2 bits (round 6): the plain and the act-order decode forms, up to 4 rows
    if (L.g_idx != nullptr && !(L.perm && L.qweight_seq)) return false;           // act-order: only with the re-sequenced rows (the copy is made of them) and perm
    if (L.dtype != GPTQ_F16 && L.dtype != GPTQ_BF16) return false;
    if (L.K % 32 || L.N % GPTQ_STRIP_COLS) return false;
    const int kpl = tiled_kpl(L.bits);
    const int gu = L.group_size / kpl;                                            // a lane's k lie in one group
    return L.group_size % kpl == 0 && (L.group_size >= L.K || (gu & (gu - 1)) == 0);
}

size_t tiled_weight_bytes(const gptq_layer_t& L) {
    const int cke = 4 * tiled_kpl(L.bits);
    return (size_t)((L.K + cke - 1) / cke) * tiled_chunk_bytes(L.bits) * (size_t)(L.N / GPTQ_STRIP_COLS);
}
size_t tiled_const_bytes(const gptq_layer_t& L) {
    return (size_t)((L.K + L.group_size - 1) / L.group_size) * tiled_rec_bytes(L.bits) * (size_t)(L.N / GPTQ_STRIP_COLS);
}

TiledPlan plan_tiled(const gptq_layer_t* const* Ls, int n, int M, const gptq_tuning_t* tune) {
    TiledPlan pl{};
    if (n < 1 || n > 4 || M < 1 || M > 8) return pl;
    const gptq_layer_t& A = *Ls[0];
    if (A.bits == 2 && M > 4) return pl;                                          // 2 bits: no 5..8-row form
    int strips = 0, nsum = 0;
    const bool pair = A.epilogue == GPTQ_EPI_SILU_MUL;                             // one [gate | up] layer: a workgroup per PAIR of strips, M <= 4, no K slices
    if (pair && (n != 1 || M > 4)) return pl;
    for (int i = 0; i < n; ++i) {
        const gptq_layer_t& L = *Ls[i];
        if (!tiled_layer_ok(L) || (L.epilogue != GPTQ_EPI_NONE) != pair) return pl;
        if (L.K != A.K || L.group_size != A.group_size || L.dtype != A.dtype || L.bits != A.bits || (L.g_idx != nullptr) != (A.g_idx != nullptr)) return pl;
        strips += L.N / GPTQ_STRIP_COLS;
        nsum += L.N;
    }
    if (pair) strips /= 2;
    pl.pair = pair;
    // Strips per workgroup (gemv_tiled_multi.hip): 3..8 rows of plain layers whose widths are whole groups of strips and that need no K slices -- every strip
    // staging its own rows of x is what those forms lose on layers of many strips.  tuning.reserved[GPTQ_LAB_OPT] = 1 / 2 / 4 (with path = 8) forces the count.
    int nstr = 1;
    const int multi_from = A.bits == 4 ? 1 : 3;       // 4-bit layers: also at 1 - 2 rows (two strips only)
    if (!pair && M >= multi_from && !A.g_idx && A.bits != 2) {
        // measured (tools/multi_strip_ab.py, profiles/r05_multi_strip_ab.log, us, 1 / 2 / 4 strips per workgroup): gate|up M = 4 13.2 / 12.6 / 16.5, M = 8 21.2 / 16.5 / 21.4
        // (the batched-decode kernel on the checkpoint rows: 17.3); q|k|v M = 8 12.8 / 11.2 / 11.3 (10.8); 4096x11008 M = 8 12.2 / 11.0 / 11.0 (10.5); 3..4 rows
        // on launches below ~1000 strips lose 0.7 - 1 us -- two strips per workgroup from 1024 strips up, else one
        // Round 6 (tools/multi_geom_sweep.py, profiles/r06_multi_geom_sweep.log): also from 768 strips where a strip's chunks do not divide into passes of the
        // one-strip form (4 waves x 4 chunks = 16 per pass) but do into the two-strip form's (2 waves x 4 = 8): K = 5120 is 40 chunks = 2.5 passes of 16 --
        // 13B q|k|v (960 strips) M = 1 / 2 / 4: 12.3 / 12.6 / 14.1 -> 10.6 / 11.4 / 13.0 us, 5120x13824 (864) 11.2 -> 10.2; K = 4096 (32 chunks: whole passes either
        // way) LOSES 7 % with two strips below 1024 (q|k|v 7B, 4096x12288) and keeps one
        const int cke0 = 4 * tiled_kpl(A.bits), chunks0 = (A.K + cke0 - 1) / cke0;
        const double waste1 = (double)((chunks0 + 15) / 16 * 16) / chunks0, waste2 = (double)((chunks0 + 7) / 8 * 8) / chunks0;
        const bool uneven = strips >= 768 && M <= 4 && (A.bits == 4 || (A.bits == 3 && M >= 3)) && waste1 - waste2 >= 0.1;      // (3 bits: the same 128-k chunks; its two-strip forms exist from 3 rows: 5120x13824 int3 M = 4 13.4 -> 11.2 us)
        const int want = (tune && tune->path == 8 && tune->reserved[1]) ? tune->reserved[1] : ((strips >= 1024 || uneven) ? 2 : 1);
        if (want == 2 || (want == 4 && M >= 3)) {
            nstr = want;
            for (int i = 0; i < n; ++i)
                if (Ls[i]->N % (GPTQ_STRIP_COLS * want) != 0) nstr = 1;
        }
    }
    pl.nseg = n;
    pl.mt = M >= 5 ? 8 : (M >= 3 ? 4 : M);                                        // 5..8 rows: a second A operand (rows 4..7), two matrix-core steps per decoded pair
    const int cke = 4 * tiled_kpl(A.bits), rec = tiled_rec_bytes(A.bits);         // k per chunk; bytes of one group's constants
    const int chunks = (A.K + cke - 1) / cke;
    pl.bits = A.bits;
    pl.chunks_total = chunks;
    pl.strips_total = strips;
    pl.nsum = nsum;
    pl.groups = (A.K + A.group_size - 1) / A.group_size;
    // K slices: narrow layers (TP shards: fewer than 128 strips) fill the chip with them, and a slice's part of x (MT rows) has to fit the LDS next to
    // the constants -- combined inside the launch through granules (stream_finish)
    int ks = (tune && tune->ksplit) ? tune->ksplit : 0;
    if (!ks) {
        // Late round 6 (tools/tp_shard_sweep.py, profiles/r06_tp_shard_sweep.log: BASELINE config 4's shards): the exchange hop of a K slice costs more than the idle CUs
        // of a narrow launch until the slices are deep -- 8192x1024 (64 strips) M = 1: 2 slices 5.69 us, none 5.31 (4.88 as 8 waves x 4 chunks), 8192x128 5.37 -> 4.73,
        // 4096x1376 5.03 -> 3.77, 4096x512 4.75 -> 3.61; 28672x1024 wants FOUR (11.45 / 9.46 / 8.15 / 12.1 us with 1 / 2 / 4 / 8): slices of at least 40 chunks = 5120 k
        ks = 1;
        while (!pair && strips * ks < 256 && chunks / (ks * 2) >= 40) ks *= 2;
    }
    if (pair && ks != 1) return pl;
    if (nstr > 1 && (ks != 1 || strips / nstr < 96)) nstr = 1;                    // K slices (narrow layers): one strip per workgroup, as before
    if (nstr > 1 && (size_t)pl.mt * ((size_t)chunks * cke * 2 + 16) + (size_t)nstr * (((size_t)pl.groups * rec + 15) & ~(size_t)15) + 16 * (pl.mt * 16 + 4) * sizeof(float) + 16 > 96 * 1024)
        nstr = 1;                                                                 // the rows of x over the whole K do not fit next to the constants (8 rows at K = 11008): K slices, one strip per workgroup
    pl.nstr = nstr;
    if (nstr > 1) strips /= nstr;
    pl.strips_total = strips;
    const size_t raw_bytes = A.g_idx ? (size_t)pl.mt * ((size_t)A.K * 2 + 16) + 16 : 0;      // act-order: does not shrink with K slices
    const size_t lds_cap = 96 * 1024 + raw_bytes;
    auto lds_need = [&](int k_slices, int waves) {
        const int cps = (chunks + k_slices - 1) / k_slices;
        return (size_t)pl.mt * ((size_t)cps * cke * 2 + 16) + (pair ? 2 : nstr) * (((size_t)pl.groups * rec + 15) & ~(size_t)15) + (size_t)waves * (pl.mt * 16 + 4) * sizeof(float) + 16 +
               (A.dtype == GPTQ_BF16 && pl.mt <= 2 ? (size_t)cps * 64 : 0) +
               (A.g_idx ? (size_t)pl.mt * ((size_t)A.K * 2 + 16) + 16 : 0);         // act-order: the raw x rows, whole K                      // bf16: one inverse block factor per (run of a chunk's 4, row of 4)
    };
    while (!pair && ks < 8 && ks < chunks && lds_need(ks, 16) > lds_cap) ks *= 2;
    if (pair && lds_need(1, 16) > lds_cap) return pl;
    if (ks > chunks) ks = chunks;
    if (ks > 8) ks = 8;                                                           // the owner's poll is unrolled over at most 7 other slices
    const int cps = (chunks + ks - 1) / ks;
    pl.chunks_per_split = cps;
    pl.ksplit = (chunks + cps - 1) / cps;                                         // no empty slices
    if ((size_t)strips * 4 > WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES - WS_HEADER_EPOCH_OFFSET) return pl;   // one epoch word per strip
    int waves, u;
    if (tune && tune->waves && tune->reserved[0]) {
        waves = tune->waves; u = tune->reserved[0];
    } else {
        // tools/tiled_sweep.py on MI355X (profiles/r04_tiled_sweep_*.log, us per launch, rotating HBM-cold layers in a hipGraph): one 16-wave workgroup per CU
        // with 2 chunks per wave in flight for the <= 256-strip launches (4096^2 4.43, 11008x4096 7.41; 8 waves x 4: 4.50 / 7.68), 4 waves x 4 chunks where
        // several workgroups share a CU (4096x11008 7.13, q|k|v 7.50, gate|up 11.11; 8 waves x 4: 7.69 / 8.21 / 12.44)
        // Round 6 (tools/strips_geom_sweep.py, profiles/r06_strips_geom_sweep*.log: the 13B / 30B / 70B shapes): 257 .. 512 workgroups as 8 waves x 2 chunks -- two
        // co-resident workgroups per CU, ONE round -- where 16-wave workgroups ran a second round for the last few strips (5120^2, 320 strips, M = 1 / 4:
        // 9.06 / 10.66 -> 6.14 / 7.43 us; 13824x5120 14.7 -> 11.8) and 4-wave ones lost 2 - 12 % (6656^2 9.06 -> 8.00, 7168^2 8.84 -> 8.14, 17920x6656 16.4 -> 15.4,
        // 4096x6144 5.77 -> 5.54; 512 workgroups: 8192^2 / 4096x8192 / 28672x8192 equal or + 3 %); up to 256 (one per CU) and from 513 on nothing changes
        const int wgs = strips * pl.ksplit;
        if (wgs <= 256) { waves = 16; u = 2; }
        else if (wgs <= 512) { waves = 8; u = 2; }
        else { waves = 4; u = 4; }
        if (wgs <= 128 && chunks >= 64 && pl.ksplit == 1) { waves = 8; u = 4; }   // narrow launches of K >= 8192 (the shards above): 8 waves x 4 chunks 4.88 against 5.31 us
        if (pl.mt > 4) { waves = 8; u = 4; }                                      // 5..8 rows, 4096^2: 6.62 us (4 x 4: 6.82, 16 x 2: 6.92)
        if (pair && waves == 4) waves = 8;                                        // two strips per workgroup: the 4-wave form's work per wave
        // the pair form where a strip's chunks do not divide into its passes of 16 (4 waves per half x 4 chunks) but do into passes of 8: 2 chunks in flight (tools/pair_geom_sweep.py,
        // profiles/r06_pair_geom_sweep.log, M = 1 / 2 / 4: 13B 5120 -> 2 x 13824 19.5 / 20.1 / 21.3 -> 18.1 / 17.5 / 18.6 us, 30B 6656 -> 2 x 17920 29.5 / 29.5 / 34.0 -> 26.0 / 27.1 / 31.6;
        // K = 4096 / 8192 divide evenly and keep 4)
        if (pair && waves == 8 && u == 4 && (double)((chunks + 15) / 16 * 16) / chunks - (double)((chunks + 7) / 8 * 8) / chunks >= 0.1) u = 2;
        if (nstr > 1) { waves = nstr == 4 ? 16 : 8; u = 4; }                      // four (two) strips x four waves x four chunks in flight
        // 1 - 2 rows, two strips: TWO waves per strip up to K = 7168 (profiles/r06_strips_geom_sweep3.log, 8 -> 4 waves: 4096x22016 11.7 -> 11.5, 5120x27648 19.2 -> 16.9,
        // 6656x17920 M = 2 18.9 -> 16.9, 4096x16384 9.7 -> 9.3 us; 8192x28672 24.4 -> 25.5: deeper layers keep four)
        if (nstr == 2 && pl.mt <= 2 && A.K <= 7168) waves = 4;
        // two strips at 3 - 4 rows where a strip's chunks divide into passes of 8 but not of 16 (4 waves per strip): 2 chunks in flight -- 5120x13824 M = 4 12.6 -> 10.8 us
        // (profiles/r06_strips_geom_sweep4.log)
        if (nstr == 2 && pl.mt == 4 && waves == 8 && (double)((chunks + 15) / 16 * 16) / chunks - (double)((chunks + 7) / 8 * 8) / chunks >= 0.1) u = 2;
        // 3 - 4 rows of a deep layer, one strip per workgroup: the staged x (4 rows x K) is what limits the workgroups of a CU -- 4-wave workgroups of K = 6656+
        // leave it at 8 waves (two workgroups in 160 KiB); 8-wave workgroups keep 16 (profiles/r06_multi_geom_sweep.log, 4 -> 8 waves: 70B q|k|v 15.7 -> 13.3 us,
        // 30B q|k|v 23.2 -> 20.4; K = 5120 holds three 4-wave workgroups and LOSES with 8 waves: 14.1 -> 16.0)
        if (pl.mt == 4 && waves == 4 && !pair && nstr == 1 && ((size_t)160 * 1024 / lds_need(ks, 4)) * 4 < 12) waves = 8;
        const int per = pair ? 2 : nstr;
        while (waves > per && (waves / (2 * per)) * u >= cps) waves /= 2;
    }
    if (pair && (waves & 1)) return pl;
    if (nstr > 1 && pl.mt <= 2 && (waves > 8 || u != 4)) return pl;              // the 1 - 2-row two-strip form is compiled for its planned geometry only
    if (nstr > 1 && waves % nstr != 0) return pl;
    if (waves < 1 || waves > 16 || (u != 2 && u != 4)) return pl;                // 2 or 4 chunks per wave in flight (the 1- / 8-chunk forms were lab-only: retired in round 6)
    if (waves > tiled_max_waves(u, nstr)) return pl;                              // 4 chunks in flight: workgroups of at most 8 waves (one compilation per depth; only a forced geometry gets here)
    pl.waves = waves;
    pl.u = u;
    pl.zm2 = A.bits == 4 && A.dtype == GPTQ_BF16 && pl.mt == 2 && (long)strips * pl.ksplit < 1024;      // profiles/r06_zm_ab.log
    pl.xstride = cps * cke * 2 + 16;
    pl.lds_bytes = lds_need(pl.ksplit, waves);
    pl.xraw_off = A.g_idx ? (int)((pl.lds_bytes - ((size_t)pl.mt * ((size_t)A.K * 2 + 16) + 16) + 15) & ~(size_t)15) : 0;
    if (pl.lds_bytes > 160 * 1024) return pl;
    pl.partial_bytes = pl.ksplit > 1 ? (size_t)(pl.ksplit - 1) * M * nsum * 8 : 0;
    pl.ok = true;
    return pl;
}

hipError_t launch_tiled(const gptq_layer_t* const* Ls, const TiledPlan& pl, const void* x, void* const* outs, int M, void* ws_header, void* ws_body,
                        hipStream_t st, const gptq_peer_group_t* pg) {
    if (!pl.ok) return hipErrorInvalidValue;
    TiledParams p{};
    if (pg) {                                                                     // one layer = this rank's column shard
        if (pl.nseg != 1 || pl.mt > 4 || pg->world < 1 || pg->world > GPTQ_PEER_MAX) return hipErrorInvalidValue;
        for (int r = 0; r < pg->world; ++r) {
            p.peer.xbuf[0][r] = (char*)pg->xbuf[0][r];
            p.peer.xbuf[1][r] = (char*)pg->xbuf[1][r];
            p.peer.flags[r] = pg->flags[r];
        }
        p.peer.state = pg->state;
        p.peer.world = pg->world;
        p.peer.rank = pg->rank;
        p.peer.row_bytes = (unsigned)pg->N * 2u;
        p.peer.col_off_bytes = (unsigned)pg->rank * (unsigned)Ls[0]->N * 2u;
        p.peer.owners = (unsigned)pl.strips_total;
    }
    int blk = 0, col = 0;
    for (int i = 0; i < 4; ++i) p.blk_end[i] = 0x7fffffff;
    for (int i = 0; i < pl.nseg; ++i) {
        const gptq_layer_t& L = *Ls[i];
        blk += L.N / GPTQ_STRIP_COLS / (pl.pair ? 2 : pl.nstr);
        p.blk_end[i] = blk;
        p.seg[i] = TiledSeg{L.qweight_tiled, L.qconst_tiled, L.bias, outs[i], L.N, col, L.g_idx ? L.perm : nullptr};
        col += L.N;
    }
    p.blk_end[3] = 0x7fffffff;                                                    // the selector adds three compares: a fourth layer is reached by the first three
    const gptq_layer_t& A = *Ls[0];
    p.x = x;
    p.gran = (unsigned long long*)ws_body;
    p.epochs = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_EPOCH_OFFSET) : nullptr;
    p.err = ws_header ? (unsigned*)((char*)ws_header + WS_HEADER_BYTES - WS_HEADER_TAIL_BYTES) + 2 : nullptr;
    p.max_spins = 1u << 20;
    p.nseg = pl.nseg; p.M = M; p.K = A.K;
    p.chunks = pl.chunks_total; p.chunks_per_split = pl.chunks_per_split; p.ksplit = pl.ksplit;
    p.gu_shift = A.group_size >= A.K ? 26 : __builtin_ctz((unsigned)(A.group_size / tiled_kpl(A.bits)));   // group_size == K: every k maps to group 0
    p.nsum = pl.nsum;
    p.groups = pl.groups;
    p.xstride = pl.xstride;
    p.waves = pl.waves;
    p.xraw_off = pl.xraw_off;
    if (pg) {                                                                     // plain layers, 2 or 4 chunks per wave in flight (the compiled forms)
        if (A.g_idx || A.bits == 2 || (pl.u != 2 && pl.u != 4)) return hipErrorInvalidValue;
        return launch_tiled_peer(pl, p, A.dtype, st);                             // gemv_tiled_peer.hip
    }
    if (pl.pair) return (pl.u == 2 || pl.u == 4) ? launch_tiled_pair(pl, p, A.dtype, st) : hipErrorInvalidValue;   // gemv_tiled_pair.hip
    if (pl.nstr > 1) return (pl.u == 2 || pl.u == 4) ? launch_tiled_multi(pl, p, A.dtype, st) : hipErrorInvalidValue;  // gemv_tiled_multi.hip
    if (A.g_idx) return launch_tiled_act(pl, p, A.dtype, st);                     // gemv_tiled_act.hip
    return A.dtype == GPTQ_BF16 ? launch_tiled_bits<bf16, 0>(pl, p, st) : launch_tiled_bits<f16, 0>(pl, p, st);
}

hipError_t init_gemv_tiled_device() {
    hipError_t e = grant_tiled_lds<0>();
    hipError_t e2 = init_gemv_tiled_act_device();
    hipError_t e3 = init_gemv_tiled_peer_device();
    hipError_t e4 = init_gemv_tiled_pair_device();
    hipError_t e5 = init_gemv_tiled_multi_device();
    return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : (e4 != hipSuccess ? e4 : e5)));
}

}  // namespace gptq
