#!/bin/bash
# Lab: tools/lab/libgptq_tiled_noxs.so = the library with the decode-copy kernel's bf16 run-sum form compiled out (-DGPTQ_TILED_NO_XS: the round-4 bf16 forms),
# for same-session A/B runs of tools/bf16_vs_f16.py under GPTQ_MI355X_LIB.  The planner's LDS sizing is shared (it reserves the run table either way).
set -e
cd "$(dirname "$0")/../../autogptq_amd/csrc"
FLAGS="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -fno-strict-aliasing -DGPTQ_TILED_NO_XS=1"
OBJS=""
for f in gemv_tiled gemv_tiled_act gemv_tiled_peer gemv_tiled_pair gemv_tiled_multi; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/${f}_noxs.o &
  OBJS="$OBJS /tmp/${f}_noxs.o"
done
wait
OTHERS=$(ls *.o | grep -v '^gemv_tiled' | grep -v gemm_strips.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/libgptq_tiled_noxs.so $OBJS $OTHERS
