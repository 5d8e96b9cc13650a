#!/bin/bash
# Lab (no product path): builds tools/lab/libgptq_wsk_abl<N>.so = the library with the stream-K prefill kernel's ablation switches compiled in
# (GPTQ_WSK_ABL bit 0: no vmcnt wait at the step end, 1: no barrier, 2: no x DMAs, 3: no dequant math -- results are WRONG by construction, timing only),
# for tools/wide_sk_ab.py under GPTQ_MI355X_LIB.  Run on the build container (cross-compiles), the .so files travel with the gpurun snapshot.
set -e
cd "$(dirname "$0")/../../autogptq_amd/csrc"
FLAGS="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -fno-strict-aliasing"
OTHERS=$(ls *.o | grep -v '^gemm_wide_sk.o$' | grep -v gemm_strips.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DGPTQ_WSK_ABL=$n -c gemm_wide_sk.hip -o /tmp/gemm_wide_sk_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/libgptq_wsk_abl$n.so /tmp/gemm_wide_sk_abl$n.o $OTHERS
done
