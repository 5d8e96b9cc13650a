mkdir -p gpurun_out/r05rows
timeout 1200 python -m pytest tests/test_gpu_rows.py -x -q -k "forced" > gpurun_out/r05rows/tests5.log 2>&1; tail -4 gpurun_out/r05rows/tests5.log
timeout 500 python tools/rows_ab.py --ms 64,96,128 --geoms 0x0,2x2,2x3,2x4,2x6,4x1,4x2,4x3,4x4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05rows/ab5.log; cat gpurun_out/r05rows/ab5.log
