#!/bin/bash
# usage (GPU box): tools/session_final2.sh <tag>  -- whole GPU test suite -> default bench line -> rocprofv3 kernel trace of the same command
# (the PMC passes of tools/prof_bench.sh are a separate session: tools/session_final.sh)
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 420 python -m pytest tests -m gpu -q --maxfail 25 --timeout 150 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 240 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1
echo "bench rc=$?"
grep '^{' $OUT/bench.log > $OUT/bench.json
cut -c1-300 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/$TAG.trace.log 2>&1
echo "trace rc=$?"
python $R/tools/rocprof_summary.py /tmp/$TAG/trace/t_results.db --match gptq --top 48 > $OUT/kernel_stats.txt
grep '^{' /tmp/$TAG.trace.log > $OUT/bench_under_trace.json
cut -c1-190 $OUT/kernel_stats.txt | head -40
