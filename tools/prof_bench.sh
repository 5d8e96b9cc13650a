#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]   (run on the GPU box)
# Separate passes, as the profiling guide prescribes: kernel trace (+stats), then FETCH_SIZE, then WRITE_SIZE counters.
# Only the text/json summaries are kept under gpurun_out/<tag>/ (the rocpd databases are tens of MB and stay in /tmp).
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/$TAG.trace.log 2>&1
python $R/tools/rocprof_summary.py /tmp/$TAG/trace/t_results.db --match gptq --top 40 > $OUT/kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/$TAG/fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/$TAG.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/$TAG/write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/$TAG.write.log 2>&1
python $R/tools/pmc_traffic.py --fetch /tmp/$TAG/fetch/p_results.db --write /tmp/$TAG/write/p_results.db --prefill-m 2048 --out $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python $R/tools/rocprof_summary.py /tmp/$TAG/fetch/p_results.db --match gptq > $OUT/pmc_fetch_summary.txt
grep '^{' /tmp/$TAG.trace.log > $OUT/bench_under_trace.json
cut -c1-200 $OUT/kernel_stats.txt
