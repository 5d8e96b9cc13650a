# SQ counters of the batched-decode kernel (csrc/gemm_rows.hip) on one shape per database: 4096^2 at M = 16 and M = 64 (default plans)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05rows
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/rows_pmc.txt
for M in 16 64; do
  CMD="python $R/tools/prefill_one.py --k 4096 --n 4096 --m $M --layers 8 --reps 20"
  timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/rp/a_$M -o p -- $CMD > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE -d /tmp/rp/b_$M -o p -- $CMD > /dev/null 2>&1
  timeout 150 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d /tmp/rp/c_$M -o p -- $CMD > /dev/null 2>&1
  echo "## 4096x4096 M=$M" >> $OUT/rows_pmc.txt
  for d in a b c; do python $R/tools/rocprof_summary.py /tmp/rp/${d}_$M/p_results.db --match rows >> $OUT/rows_pmc.txt 2>&1; done
done
cut -c1-170 $OUT/rows_pmc.txt
