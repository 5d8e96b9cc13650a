#!/bin/bash
# round 6, first session of the panel kernel: forced-geometry parity, then the A/B sweep against the default plan
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_panel.py -x -q -k forced 2>&1 | tail -15 > gpurun_out/r06_panel_tests1.log
cat gpurun_out/r06_panel_tests1.log
timeout 1200 python tools/panel_ab.py --ms 128,256,512,768 --shapes 4096x4096,4096x11008,11008x4096 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_panel_ab1.log
cat gpurun_out/r06_panel_ab1.log
