mkdir -p gpurun_out/r05rows
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q > gpurun_out/r05rows/tests.log 2>&1; tail -3 gpurun_out/r05rows/tests.log
timeout 900 python tools/rows_ab.py --ms 5,8,16,32,64,96,128,192,256 --geoms 0x0,1x1,1x2,1x3,1x4,2x2,2x3,2x4,2x6 2>&1 | grep -v amdgpu.ids > gpurun_out/r05rows/ab.log; cat gpurun_out/r05rows/ab.log
