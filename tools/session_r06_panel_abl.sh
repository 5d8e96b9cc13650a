#!/bin/bash
# round 6: what each part of the panel kernel's K loop costs -- the product against builds with parts compiled out (tools/ab_unit.sh ablN gemm_panel.hip -DGPTQ_PANEL_ABL=N:
# 1 no vmcnt wait, 2 no x DMAs, 4 no weight loads, 8 no dequant math; timing only, results are wrong by construction)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="--ms 256,512 --shapes 4096x4096 --geoms 22x8,24x8,22x4,24x4 --check 0 --rounds 2"
for lib in product abl1 abl2 abl4 abl8 abl3 abl7 abl15 product; do
  if [ "$lib" = product ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so; fi
  echo "== lib $lib"
  timeout 600 python tools/panel_ab.py $ARGS 2>&1 | grep -v amdgpu.ids
done
