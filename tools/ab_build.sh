#!/bin/bash
# usage (build container): tools/ab_build.sh <NAME> <file>[=gemv.hip|gemm.hip|utils.hip|capi.hip] [git-ref]
#   tools/ab_build.sh B /tmp/my_gemv.hip                 -> tools/libgptq_B.so with gemv.hip replaced by that file
#   tools/ab_build.sh C gemm.hip HEAD~3                  -> ... with gemm.hip taken from that commit
# Links the replaced translation unit against the CURRENT objects of the other three (run `make -C autogptq_amd/csrc` first).
# The .so is git-ignored and travels to the GPU box with the snapshot; select it with GPTQ_MI355X_LIB (tools/ab_run.sh).
set -eu
NAME=$1; SRC=$2; REF=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/autogptq_amd/csrc
W=$(mktemp -d)/a/b; mkdir -p "$W"                       # two levels deep: the sources include "../../include/gptq_mi355x.h"
mkdir -p "$W/../../include"; cp "$ROOT/include/gptq_mi355x.h" "$W/../../include/"
cp "$CS/common.cuh" "$CS/launch.h" "$W/"
if [ -n "$REF" ]; then UNIT=$(basename "$SRC"); git -C "$ROOT" show "$REF:autogptq_amd/csrc/$UNIT" > "$W/$UNIT"
else UNIT=gemv.hip; case "$(basename "$SRC")" in gemm_mid*.hip) UNIT=gemm_mid.hip;; gemm*.hip) UNIT=gemm.hip;; utils*.hip) UNIT=utils.hip;; capi*.hip) UNIT=capi.hip;; esac; cp "$SRC" "$W/$UNIT"; fi
OBJ=${UNIT%.hip}.o
(cd "$W" && /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -fno-strict-aliasing -c "$UNIT" -o "$OBJ")
OTHERS=""; for o in capi.o gemv.o gemm.o utils.o peer.o mlp.o gemm_mid.o; do [ "$o" = "$OBJ" ] && OTHERS="$OTHERS $W/$OBJ" || OTHERS="$OTHERS $CS/$o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libgptq_$NAME.so" $OTHERS
ls -la "$ROOT/tools/libgptq_$NAME.so"
