#!/bin/bash
# usage (GPU box): tools/session_r04_prof.sh <tag>  -- the round's rocprof evidence: kernel trace + stats of the default bench command, the two PMC traffic
# passes (FETCH_SIZE, WRITE_SIZE: separate runs), and the SQ counters of the prefill kernels (M = 4096 wide tile, M = 2048 desc_act).
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_bench.sh $TAG > /dev/null 2>&1
OUT=$R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
# one rocprofv3 run per shape and counter set (a second process under the same -o overwrites the first's database)
: > $OUT/gemm_pmc.txt
i=0
for ARGS in "--ms 4096 --shapes 4096x4096" "--ms 2048 --shapes 4096x11008"; do
  i=$((i+1))
  CMD="python $R/tools/prefill_shapes.py $ARGS"
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/$TAG/g1_$i -o p -- $CMD > $OUT/gemm_pmc1_$i.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d /tmp/$TAG/g2_$i -o p -- $CMD > $OUT/gemm_pmc2_$i.log 2>&1
  echo "## $ARGS" >> $OUT/gemm_pmc.txt
  python $R/tools/rocprof_summary.py /tmp/$TAG/g1_$i/p_results.db --match gemm >> $OUT/gemm_pmc.txt 2>&1
  python $R/tools/rocprof_summary.py /tmp/$TAG/g2_$i/p_results.db --match gemm >> $OUT/gemm_pmc.txt 2>&1
done
cut -c1-220 $OUT/kernel_stats.txt | head -30
cat $OUT/pmc_traffic.txt | cut -c1-300 | head -20
