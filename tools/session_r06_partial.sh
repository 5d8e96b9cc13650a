#!/bin/bash
# round 6: partial single panels (33 .. 63 rows) -- parity of the panel tests, then the full-tile path A/B against the library without the clamp (tools/libgptq_NOPART.so),
# then the default plans across the affected row counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r06_partial_panel.log
timeout 900 python -m pytest tests/test_gpu_panel.py -m gpu -x -q 2>&1 | tail -5 > $OUT
for r in 1 2; do for v in A NOPART; do
  if [ "$v" = A ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$v.so; fi
  echo "### library $v round $r" >> $OUT
  timeout 300 python tools/m_sweep.py --ms 64,128,192,256,384,512 --shapes 4096x4096,4096x11008 2>&1 | grep -v amdgpu >> $OUT
done; done
unset GPTQ_MI355X_LIB
echo "### default plans, 24 .. 96 rows" >> $OUT
timeout 300 python tools/m_sweep.py --ms 24,32,33,40,48,56,63,64,80 --shapes 4096x11008,4096x4096 2>&1 | grep -v amdgpu >> $OUT
timeout 300 python tools/m_sweep.py --bits 3 --gs 32 --ms 32,33,48,63,64 --shapes 4096x11008 2>&1 | grep -v amdgpu >> $OUT
timeout 300 python tools/m_sweep.py --bits 8 --gs 32 --ms 32,33,48,63,64 --shapes 4096x11008 2>&1 | grep -v amdgpu >> $OUT
timeout 300 python tools/m_sweep.py --act --ms 32,33,48,63,64 --shapes 4096x11008 2>&1 | grep -v amdgpu >> $OUT
cat $OUT
