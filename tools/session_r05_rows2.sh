# rows kernel: (1) the cross-wave sum with four loads in flight, 7B shapes; (2) other model shapes, old default (GPTQ_LAB_NO_ROWS=1) against the new default
mkdir -p gpurun_out/r05rows
timeout 600 python -m pytest tests/test_gpu_rows.py -x -q -k "fp16" > gpurun_out/r05rows/tests2.log 2>&1; tail -2 gpurun_out/r05rows/tests2.log
for mode in old new; do
  if [ $mode = old ]; then export GPTQ_LAB_NO_ROWS=1; else unset GPTQ_LAB_NO_ROWS; fi
  timeout 500 python tools/rows_ab.py --ms 8,16,32,64,128 --geoms 0x0 --shapes 4096x4096,4096x11008,11008x4096,5120x5120,5120x13824,13824x5120,8192x8192,8192x28672,28672x8192,2048x2048,4096x2048,8192x1024,1024x8192 2>&1 | grep -v amdgpu.ids | sed "s/^/[$mode default] /" >> gpurun_out/r05rows/ab_shapes.log
done
cat gpurun_out/r05rows/ab_shapes.log
