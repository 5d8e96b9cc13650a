#!/bin/bash
# usage (GPU box): tools/session_tests_ab.sh <tag>   -- whole GPU test suite, then the 3- / 8-bit decode A/B (tools/magic_ab.py)
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests -m gpu -q --maxfail 25 --timeout 150 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
timeout 200 python tools/magic_ab.py > $OUT/magic_ab.log 2>&1
echo "magic_ab rc=$?" >> $OUT/magic_ab.log
tail -15 $OUT/pytest.log
cat $OUT/magic_ab.log | grep -v amdgpu.ids
