#!/bin/bash
# usage (GPU box): tools/session_nonq4.sh <tag>  -- 3- / 8-bit parity tests (GEMV + GEMM paths, act-order on / off), then tools/nonq4_paths.py
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -m gpu -q --maxfail 25 --timeout 150 -p no:cacheprovider \
    -k "config5 or int3 or int8 or bits or magic or fixture or random_words or gemm or reference_backend_grid or dequant" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 200 python tools/nonq4_paths.py > $OUT/nonq4_paths.log 2>&1
echo "nonq4 rc=$?" >> $OUT/nonq4_paths.log
grep -v amdgpu.ids $OUT/nonq4_paths.log
