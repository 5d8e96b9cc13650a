#!/bin/bash
# round 6: the whole GPU suite, then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gpu_tests.log
echo "pytest rc=${PIPESTATUS[0]}" >> gpurun_out/r06_gpu_tests.log
cat gpurun_out/r06_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -c 3000 gpurun_out/r06_bench.json
