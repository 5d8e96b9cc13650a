#!/bin/bash
# usage (GPU box): tools/session_r06_prof.sh <tag>  -- the round's rocprof evidence (every rocprofv3 run under `timeout`; counter passes are --pmc only):
#   (1) kernel trace + stats and the FETCH_SIZE / WRITE_SIZE passes of the default bench command (tools/prof_bench.sh: decode kernels, keyed by grid);
#   (2) the panel kernel (csrc/gemm_panel.hip) ONE SHAPE PER DATABASE: kernel trace, FETCH_SIZE / WRITE_SIZE -> rows appended to pmc_traffic.json with their
#       (K, N, M), and the SQ counter sets on M = 512 / 256 of 4096 -> 4096; the config-3 layer and north_star's M = 4096 as in round 5.
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 bash $R/tools/prof_bench.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
: > $OUT/panel_pmc.txt
: > $OUT/panel_kernel_stats.txt
: > $OUT/prefill_one.log
i=0
for SPEC in "4096 4096 512" "4096 4096 256" "4096 11008 128" "11008 4096 512" "4096 11008 2048 --act" "4096 4096 4096"; do
  i=$((i+1))
  set -- $SPEC
  K=$1; N=$2; M=$3; ACT="${4:-}"
  CMD="python $R/tools/prefill_one.py --k $K --n $N --m $M $ACT --layers 8 --reps 6"
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/$TAG/kt_$i -o t -- $CMD >> $OUT/prefill_one.log 2>&1
  echo "## K=$K N=$N M=$M $ACT" >> $OUT/panel_kernel_stats.txt
  python $R/tools/rocprof_summary.py /tmp/$TAG/kt_$i/t_results.db --match gptq --top 6 >> $OUT/panel_kernel_stats.txt 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE -d /tmp/$TAG/pf_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE -d /tmp/$TAG/pw_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
  python $R/tools/pmc_traffic.py --fetch /tmp/$TAG/pf_$i/p_results.db --write /tmp/$TAG/pw_$i/p_results.db --label-gemm $K,$N,$M --append $OUT/pmc_traffic.json \
         --out $OUT/pmc_traffic.json >> $OUT/pmc_traffic.txt 2>&1
  if [ $i -le 2 ]; then
    timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/$TAG/g1_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
    timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d /tmp/$TAG/g2_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
    timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d /tmp/$TAG/g3_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
    echo "## K=$K N=$N M=$M $ACT" >> $OUT/panel_pmc.txt
    for d in g1 g2 g3; do python $R/tools/rocprof_summary.py /tmp/$TAG/${d}_$i/p_results.db --match panel >> $OUT/panel_pmc.txt 2>&1; done
  fi
done
cut -c1-200 $OUT/kernel_stats.txt | head -24
cut -c1-200 $OUT/panel_kernel_stats.txt
grep -h "us per layer call" $OUT/prefill_one.log | cut -c1-120 | head -40
cut -c1-230 $OUT/panel_pmc.txt | head -30
