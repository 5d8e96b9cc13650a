#!/bin/bash
# usage (build container): tools/ab_tiled.sh <NAME> [extra hipcc flags...]
#   tools/ab_tiled.sh ZM1 -DGPTQ_TILED_ZM=1   -> tools/libgptq_ZM1.so = the current objects with the five translation units of the decode-copy kernel
# (gemv_tiled*.hip include gemv_tiled_kernel.cuh) rebuilt with the flags; select the library with GPTQ_MI355X_LIB=tools/libgptq_ZM1.so
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/autogptq_amd/csrc
UNITS="gemv_tiled gemv_tiled_act gemv_tiled_peer gemv_tiled_pair gemv_tiled_multi"
for u in $UNITS; do
  (cd "$CS" && /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -fno-strict-aliasing "$@" -c $u.hip -o /tmp/ab_${NAME}_$u.o) &
done
wait
SRCS=$(sed -n 's/^SRCS := //p' "$CS/Makefile")
OBJS=""; for s in $SRCS; do b=${s%.hip}; case " $UNITS " in *" $b "*) OBJS="$OBJS /tmp/ab_${NAME}_$b.o";; *) OBJS="$OBJS $CS/$b.o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libgptq_$NAME.so" $OBJS
ls -la "$ROOT/tools/libgptq_$NAME.so"
