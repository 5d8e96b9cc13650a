mkdir -p gpurun_out/r05rows
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q -k "several_layers" > gpurun_out/r05rows/tests3.log 2>&1; tail -3 gpurun_out/r05rows/tests3.log
rm -f gpurun_out/r05rows/ab_multi.log
GPTQ_LAB_NO_ROWS=1 timeout 400 python tools/rows_multi_ab.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05rows/ab_multi.log
timeout 400 python tools/rows_multi_ab.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05rows/ab_multi.log
cat gpurun_out/r05rows/ab_multi.log
timeout 300 python tools/rows_ab.py --ms 8,16,32,64,128 --geoms 0x0 2>&1 | grep -v amdgpu.ids > gpurun_out/r05rows/ab_auto.log; cat gpurun_out/r05rows/ab_auto.log
