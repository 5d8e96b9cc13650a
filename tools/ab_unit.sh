#!/bin/bash
# usage (build container): tools/ab_unit.sh <NAME> <unit.hip> [extra hipcc flags...]
#   tools/ab_unit.sh B gemm_panel.hip -DGPTQ_PANEL_AGPR   -> tools/libgptq_B.so = the current objects with that translation unit rebuilt with the flags
# (the Makefile's object list is used, so this follows the library as it grows; select the library with GPTQ_MI355X_LIB=tools/libgptq_B.so)
set -eu
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/autogptq_amd/csrc
OBJ=/tmp/ab_${NAME}_${UNIT%.hip}.o
(cd "$CS" && /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -fno-strict-aliasing "$@" -c "$UNIT" -o "$OBJ")
SRCS=$(sed -n 's/^SRCS := //p' "$CS/Makefile")
OTHERS=""; for s in $SRCS; do o=${s%.hip}.o; [ "$s" = "$UNIT" ] && OTHERS="$OTHERS $OBJ" || OTHERS="$OTHERS $CS/$o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libgptq_$NAME.so" $OTHERS
ls -la "$ROOT/tools/libgptq_$NAME.so"
