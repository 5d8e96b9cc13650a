#!/bin/bash
# round 6: the per-output error model (tests/test_gpu_error_model.py) -- report mode first (worst err / bound per kernel family), then asserting
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
GPTQ_ERR_MODEL_REPORT=1 timeout 900 python -m pytest tests/test_gpu_error_model.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06_error_model_report.log
timeout 900 python -m pytest tests/test_gpu_error_model.py -m gpu -q -s 2>&1 | tail -40 > gpurun_out/r06_error_model.log
cat gpurun_out/r06_error_model_report.log; tail -15 gpurun_out/r06_error_model.log
