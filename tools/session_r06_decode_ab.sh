cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for lib in product r6start; do
    if [ "$lib" = product ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so; fi
    echo "== lib $lib (rep $rep)"
    timeout 300 python tools/bf16_vs_f16.py --m 1 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r06_decode_isa_ab.log 2>&1
cat gpurun_out/r06_decode_isa_ab.log
