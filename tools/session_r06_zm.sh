#!/bin/bash
# round 6: the zero-point on the matrix core (gemv_tiled_kernel.cuh, -DGPTQ_TILED_ZM=1: bf16 layers, =2: fp16 too) against the product, same session:
# the decode-copy GPU tests under the lab library, then tools/bf16_vs_f16.py at 1 / 2 / 4 rows, interleaved, two passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LIBS=${LIBS:-"ZM1 ZM2"}
for lib in $LIBS; do
  echo "== tests under $lib"
  GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -6
done > gpurun_out/r06_zm_tests.log 2>&1
cat gpurun_out/r06_zm_tests.log
for rep in 1 2; do
  for lib in product $LIBS; do
    if [ "$lib" = product ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so; fi
    for m in 1 2 4; do
      echo "== lib $lib (rep $rep)"
      timeout 300 python tools/bf16_vs_f16.py --m $m 2>&1 | grep -v amdgpu.ids
    done
  done
done > gpurun_out/r06_zm_ab.log 2>&1
cat gpurun_out/r06_zm_ab.log
