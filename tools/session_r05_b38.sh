mkdir -p gpurun_out/r05b38
timeout 600 python -m pytest tests/test_gpu_wide_sk.py -x -q > gpurun_out/r05b38/tests.log 2>&1; tail -5 gpurun_out/r05b38/tests.log
timeout 300 python tools/wide_sk_ab.py --bits 8 --gs 32 --ms 512,1024,2048,4096 --act 0 > gpurun_out/r05b38/ab_int8_g32.log 2>&1; cat gpurun_out/r05b38/ab_int8_g32.log | grep -v amdgpu.ids
timeout 200 python tools/wide_sk_ab.py --bits 8 --gs 128 --ms 2048 --act 0,1 > gpurun_out/r05b38/ab_int8_g128.log 2>&1; cat gpurun_out/r05b38/ab_int8_g128.log | grep -v amdgpu.ids
timeout 200 python tools/wide_sk_ab.py --bits 4 --gs 128 --ms 2048 --act 0 > gpurun_out/r05b38/ab_int4_g128.log 2>&1; cat gpurun_out/r05b38/ab_int4_g128.log | grep -v amdgpu.ids
