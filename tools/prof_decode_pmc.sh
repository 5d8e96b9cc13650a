#!/bin/bash
# usage: tools/prof_decode_pmc.sh <tag>   (GPU box) -- SQ / TCP / TCC counter passes over a short decode bench (8 blocks), one file per pass
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --blocks 8 --steps 3 --warmup 1 --no-extras --no-fused --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_DEP_WAIT SQ_INST_CYCLES_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d /tmp/$TAG/p$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/$TAG/p$i/p_results.db --match gemv > $OUT/pmc$i.txt 2>&1
done
tail -n +1 $OUT/pmc*.txt | cut -c1-160 | grep -v "^#" | head -150
