#!/bin/bash
# round 6: same-session A/B of two builds of the panel kernel (product vs tools/libgptq_A.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="--ms 256,384,512 --shapes 4096x4096,4096x11008,11008x4096 --geoms 0,22,24 --check 1 --rounds 2"
for rep in 1 2; do
  for lib in product A ${EXTRA_LIBS}; do
    if [ "$lib" = product ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$lib.so; fi
    echo "== lib $lib (rep $rep)"
    timeout 600 python tools/panel_ab.py $ARGS 2>&1 | grep -v amdgpu.ids
  done
done
