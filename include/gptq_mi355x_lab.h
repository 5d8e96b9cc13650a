/* gptq_mi355x_lab.h -- names for the LAB switches carried in gptq_tuning_t.reserved[] (include/gptq_mi355x.h).
 *
 * NOT part of the drop-in boundary: a reference-side binding passes tuning = NULL.  These switches exist for the measurement tools under tools/ (sweeps,
 * interleaved A/B runs, ablations) and for the forced-geometry grids of the test suite, which pin every kernel the planner can choose.  A slot means
 * what the kernel family selected by gptq_tuning_t.path says it means; 0 is always "the planner's value".
 *
 * One switch is an ENVIRONMENT variable (read once per process): GPTQ_LAB_NO_ROWS=1 -- the planner as it was before csrc/gemm_rows.hip (the exchange-free
 * batched-decode kernel), for old-default-against-new-default runs with tuning = NULL (tools/session_r05_rows2.sh, tools/rows_multi_ab.py). */
#ifndef GPTQ_MI355X_LAB_H
#define GPTQ_MI355X_LAB_H

/* reserved[GPTQ_LAB_DEPTH]: loads in flight per lane / wave -- packed rows per lane (checkpoint-layout GEMVs), chunks per wave (decode-copy kernel),
 * K-steps per burst (batched-decode kernel), LDS stages (17..256-row kernel). */
#define GPTQ_LAB_DEPTH 0

/* reserved[GPTQ_LAB_OPT]: one option of the selected kernel family */
#define GPTQ_LAB_OPT 1
#define GPTQ_LAB_OPT_FIELD_DECODE 1   /* 3- / 8-bit fp16 matrix-core GEMV: field-by-field decode instead of the packed magic-number one */
#define GPTQ_LAB_OPT_MID_XREG 1       /* 17..256-row kernel: x through registers instead of LDS DMA */
#define GPTQ_LAB_OPT_MID_GRANULES 3   /* 17..256-row kernel: granule instead of flag combine */
#define GPTQ_LAB_OPT_GEMM_BK32 32     /* tiled GEMM: force the 32-deep K-step */

/* reserved[GPTQ_LAB_GEMM_KERNEL]: which MFMA GEMM (path = GPTQ_PATH_GEMM) */
#define GPTQ_LAB_GEMM_KERNEL 2
#define GPTQ_LAB_GEMM_SKINNY 1        /* 64-column strips, waves split K, reduce launch */
#define GPTQ_LAB_GEMM_TILED 2         /* 128 x 256 tiles */
#define GPTQ_LAB_GEMM_STRIP16 3       /* 16-column strips (4-bit, M <= 64) */
#define GPTQ_LAB_GEMM_STREAM64 4      /* batched decode: 64-column strips by LDS DMA (4-bit, M <= 64) */
#define GPTQ_LAB_GEMM_MID 5           /* 17..256-row kernel */

/* reserved[GPTQ_LAB_GEMM_VARIANT]: tiled-GEMM schedule / planner rules (A/B runs) */
#define GPTQ_LAB_GEMM_VARIANT 3
#define GPTQ_LAB_VARIANT_PLAIN_LOOP 1
#define GPTQ_LAB_VARIANT_SETPRIO 2
#define GPTQ_LAB_VARIANT_CROSS_STEP 3
#define GPTQ_LAB_VARIANT_REG_STAGED_X 5   /* act-order: register-staged x instead of LDS DMA */
#define GPTQ_LAB_VARIANT_ONE_K_GROUP 6    /* force 4-wave workgroups */
#define GPTQ_LAB_VARIANT_TWO_K_GROUPS 7   /* force 8-wave workgroups (two K halves) */
#define GPTQ_LAB_VARIANT_TAIL_ON 40       /* balanced tail: the planner's rule (the default) */
#define GPTQ_LAB_VARIANT_TAIL_OFF 41      /* whole tiles only */
#define GPTQ_LAB_VARIANT_TAIL_NO_LIMIT 42 /* the rule without its 1024-tile limit */
#define GPTQ_LAB_VARIANT_WIDE_OFF 44      /* never the 128 x 512 kernel */
#define GPTQ_LAB_VARIANT_WIDE_ON 45       /* the 128 x 512 kernel wherever it is legal */
#define GPTQ_LAB_VARIANT_WIDE_ROWS 46     /* the planner's rule, but the checkpoint rows even when the layer carries a decode copy (register-staged x) */
#define GPTQ_LAB_VARIANT_WIDE_ROWS_ON 47  /* 45 + 46 */
#define GPTQ_LAB_VARIANT_WIDE_SK_ON 48    /* the stream-K form of the wide tile (gemm_wide_sk.hip) wherever it is legal */
#define GPTQ_LAB_VARIANT_WIDE_SK_OFF 49   /* never the stream-K form */
#define GPTQ_LAB_VARIANT_ROWS_ON 50       /* the exchange-free batched-decode kernel (gemm_rows.hip) wherever it is legal; reserved[0] = row blocks of 16 (1 / 2), reserved[1] = strips per workgroup (0: the planner's) */
#define GPTQ_LAB_VARIANT_ROWS_OFF 51      /* never that kernel */
#define GPTQ_LAB_VARIANT_PANEL_ON 52      /* the whole-K panel kernel (gemm_panel.hip) wherever it is legal; reserved[0] = 20 + NT (NT = column blocks of 32 per workgroup tile, 1..4; 0: the planner's) */
#define GPTQ_LAB_VARIANT_PANEL_OFF 53     /* never that kernel */
#define GPTQ_LAB_VARIANT_MLP_TWO_PASSES 54 /* gptq_mlp_forward_ex (path = 0): SiLU * mul and the x permute of an act-order down projection as the two passes of round 5 */
/* (9..24, 32: ablation / timeline / ping-pong variants compiled only into tools/gemmlab with -DGPTQ_GEMM_ABLATIONS) */

#endif /* GPTQ_MI355X_LAB_H */
