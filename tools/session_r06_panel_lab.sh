#!/bin/bash
# round 6: same-session A/B of lab builds of the panel kernel (tools/ab_unit.sh <NAME> gemm_panel.hip -D...): product vs each library, two passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export GPTQ_LAB_NO_PANEL=1
ARGS="--ms 128,256,512 --shapes 4096x4096,4096x11008,11008x4096 --geoms 0 --check 1 --rounds 2"
for rep in 1 2; do
  for lib in product $LIBS; do
    if [ "$lib" = product ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so; fi
    echo "== lib $lib (rep $rep)"
    timeout 600 python tools/panel_ab.py $ARGS 2>&1 | grep -v amdgpu.ids | sed 's/| without.*| auto/| auto/' | cut -c1-110
  done
done
