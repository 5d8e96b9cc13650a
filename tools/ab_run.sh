#!/bin/bash
# usage (GPU box, from the repo root): tools/ab_run.sh "A B C" [rounds] [bench args...]
#   A = the in-tree library; any other name = tools/libgptq_<name>.so (tools/ab_build.sh).  Interleaved rounds, one line each:
#   variant  value  ms_per_step  fused GB/s  {per-shape us}
# Every bench run is under `timeout`: a wedged run costs 3 minutes, not the rest of the GPU budget.
set -u
VARIANTS=$1; ROUNDS=${2:-2}; shift; shift || true
for i in $(seq "$ROUNDS"); do for v in $VARIANTS; do
    if [ "$v" = A ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/libgptq_$v.so; fi
    BENCH_WATCHDOG_S=150 timeout 180 python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('$v', d['value'], d['ms_per_step'], d.get('fused_callers', {}).get('GB_per_s'), d['roofline'].get('us_per_launch_by_shape'))
except Exception as e:
    print('$v', 'FAILED', repr(e)[:80])"
done; done
