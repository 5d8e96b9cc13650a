#!/bin/bash
# round 6: after the instantiation cut (no kernel of the library touches scratch any more): the whole GPU suite, the changed fallbacks against the library before
# the cut (tools/libgptq_r6pre.so, built from the previous commit in a worktree), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_sprawl_gpu_tests.log
echo "pytest rc=${PIPESTATUS[0]}" >> gpurun_out/r06_sprawl_gpu_tests.log
cat gpurun_out/r06_sprawl_gpu_tests.log
for rep in 1 2; do
  for lib in product r6pre; do
    if [ "$lib" = product ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so; fi
    echo "== lib $lib (rep $rep)"
    timeout 600 python tools/fallback_ab.py 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r06_fallback_ab.log 2>&1
unset GPTQ_MI355X_LIB
cat gpurun_out/r06_fallback_ab.log
timeout 900 python bench.py > gpurun_out/r06_sprawl_bench.json 2> gpurun_out/r06_sprawl_bench.err; tail -c 1500 gpurun_out/r06_sprawl_bench.json
