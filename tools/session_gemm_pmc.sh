#!/bin/bash
# usage (GPU box): tools/session_gemm_pmc.sh <tag>  -- SQ / traffic counter passes (tools/prof_pmc.sh: separate rocprofv3 runs) of the product
# tiled GEMM through tools/gemmlab: M = 2048 on 4096^2 (one 8-wave two-K-group workgroup per CU) and M = 4096 (two 4-wave workgroups per CU)
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
for cfg in "2048 4096 4096" "4096 4096 4096"; do
    name=$(echo $cfg | tr ' ' 'x')
    timeout 100 $R/tools/prof_pmc.sh $TAG/$name $R/tools/gemmlab $cfg 0 2 > /dev/null 2>&1
    O=$R/gpurun_out/$TAG/$name
    {
        echo "# tools/gemmlab $cfg 0 2 (product default plan)"; grep "us " $O/trace.log | head -3
        for p in trace/t pmc1/p pmc2/p pmc3/p pmc4/p; do
            python $R/tools/rocprof_summary.py $O/${p}_results.db --match gemm_kernel --top 4 2>/dev/null | grep -v "^# source"
        done
    } > $R/gpurun_out/$TAG/gemm_pmc_$name.txt 2>&1
    rm -rf $O/trace $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
    cut -c1-200 $R/gpurun_out/$TAG/gemm_pmc_$name.txt
done
