#!/bin/bash
# usage (GPU box): tools/session_r05_prof.sh <tag>  -- the round's rocprof evidence:
#   (1) kernel trace + stats and the FETCH_SIZE / WRITE_SIZE passes of the default bench command (tools/prof_bench.sh: decode kernels, keyed by grid);
#   (2) the prefill kernels ONE SHAPE PER DATABASE (the stream-K kernel's grid is 256 workgroups whatever the shape): FETCH_SIZE, WRITE_SIZE -> rows
#       appended to pmc_traffic.json with their (K, N, M); SQ counter sets for the config-3 layer and north_star's M = 4096.
# Counter passes are separate rocprofv3 runs with --pmc only (MI355X_MICROARCH.md).
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_bench.sh $TAG > /dev/null 2>&1
OUT=$R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
: > $OUT/gemm_pmc.txt
: > $OUT/prefill_one.log
i=0
# (K N M flags): config 3 on the three shapes, north_star's M = 4096, then config 5's prefill rows (the 3- / 8-bit forms of the stream-K kernel: gemm_wide_sk_b38.hip)
for SPEC in "4096 4096 2048 --act" "4096 11008 2048 --act" "11008 4096 2048 --act" "4096 4096 4096" "4096 11008 2048 --bits=3 --gs=32" "4096 11008 2048 --bits=8 --gs=32"; do
  i=$((i+1))
  set -- $SPEC
  K=$1; N=$2; M=$3; ACT="${4:-} ${5:-}"
  CMD="python $R/tools/prefill_one.py --k $K --n $N --m $M $ACT"
  timeout 150 rocprofv3 --pmc FETCH_SIZE -d /tmp/$TAG/pf_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE -d /tmp/$TAG/pw_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
  python $R/tools/pmc_traffic.py --fetch /tmp/$TAG/pf_$i/p_results.db --write /tmp/$TAG/pw_$i/p_results.db --label-gemm $K,$N,$M --append $OUT/pmc_traffic.json \
         --out $OUT/pmc_traffic.json >> $OUT/pmc_traffic.txt 2>&1
  if [ $i -eq 2 ] || [ $i -eq 4 ] || [ $i -eq 5 ] || [ $i -eq 6 ]; then
    timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/$TAG/g1_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
    timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d /tmp/$TAG/g2_$i -o p -- $CMD >> $OUT/prefill_one.log 2>&1
    echo "## K=$K N=$N M=$M $ACT" >> $OUT/gemm_pmc.txt
    python $R/tools/rocprof_summary.py /tmp/$TAG/g1_$i/p_results.db --match gemm >> $OUT/gemm_pmc.txt 2>&1
    python $R/tools/rocprof_summary.py /tmp/$TAG/g2_$i/p_results.db --match gemm >> $OUT/gemm_pmc.txt 2>&1
  fi
done
cut -c1-220 $OUT/kernel_stats.txt | head -30
grep -h "us per layer call" $OUT/prefill_one.log | head
grep gemm $OUT/pmc_traffic.txt | cut -c1-260 | tail -14
