# the 64-row form of the batched-decode kernel at 160 .. 640 rows: forced geometries against the planner's default (older kernels / stream-K prefill kernel)
mkdir -p gpurun_out/r05rows
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q -k "forced and fp16" > gpurun_out/r05rows/tests6.log 2>&1; tail -3 gpurun_out/r05rows/tests6.log
timeout 600 python tools/rows_ab.py --ms 160,192,256,320,384,512,640 --geoms 2x4,2x6,4x2,4x3,4x4 --rounds 2 --layers 6 2>&1 | grep -v amdgpu.ids > gpurun_out/r05rows/ab6.log; cat gpurun_out/r05rows/ab6.log
