#!/bin/bash
# usage (GPU box): tools/session_gemm_lab.sh <tag>   -- the tiled-GEMM lab variants of tools/gemmlab in one short call (each run ~1 s):
# default vs loads-between-MFMA-groups (22) vs ping-pong between the K groups (24 plain, 25 act-order + DMA) vs forced one / two K groups
# (16 / 17), then the s_memtime timelines (20 / 21).  Build tools/gemmlab first (first lines of tools/gemmlab.hip; needs -DGPTQ_GEMM_ABLATIONS).
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
{
    for shape in "2048 4096 4096" "4096 4096 4096" "2048 4096 11008" "2048 11008 4096" "512 4096 4096" "128 4096 11008"; do
        timeout 20 $R/tools/gemmlab $shape 0,2,16,17,22,24,25 2
    done
    timeout 20 $R/tools/gemmlab 2048 4096 4096 21 2
    timeout 20 $R/tools/gemmlab 4096 4096 4096 20 2
} > $OUT/gemm_lab.log 2>&1
cat $OUT/gemm_lab.log
