#!/bin/bash
# usage (GPU box): tools/session_final.sh <tag>
# One session: whole GPU test suite -> 3- / 8-bit decode A/B -> the default bench line -> the three rocprofv3 passes of bench.py
# (tools/prof_bench.sh).  Most important first: a later step that runs out of time leaves the earlier outputs in gpurun_out/<tag>/.
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 420 python -m pytest tests -m gpu -q --maxfail 25 --timeout 150 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 150 python tools/magic_ab.py --rounds 2 > $OUT/magic_ab.log 2>&1
echo "magic_ab rc=$?" >> $OUT/magic_ab.log
grep -v amdgpu.ids $OUT/magic_ab.log
timeout 240 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1
echo "bench rc=$?"
grep '^{' $OUT/bench.log > $OUT/bench.json
cut -c1-400 $OUT/bench.json
timeout 420 tools/prof_bench.sh $TAG > $OUT/prof.log 2>&1
echo "prof rc=$?"
tail -25 $OUT/prof.log | cut -c1-180
