mkdir -p gpurun_out/r05rows
timeout 1200 python -m pytest tests/test_gpu_rows.py -x -q > gpurun_out/r05rows/tests4.log 2>&1; tail -3 gpurun_out/r05rows/tests4.log
timeout 400 python tools/rows_ab.py --ms 8,16,32,64,128 --geoms 0x0,1x1,1x2,2x2,2x4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05rows/ab4.log; cat gpurun_out/r05rows/ab4.log
