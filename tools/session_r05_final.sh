#!/bin/bash
# usage (GPU box): tools/session_r05_final.sh <tag> -- the round's closing evidence in ONE lease: the whole GPU suite, the default bench line, the rocprof
# session (kernel stats + FETCH / WRITE + SQ counters: tools/session_r05_prof.sh), fp16 against bf16 decode, the launch-floor context.
TAG=${1:-r05final}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests.log)
tail -4 $OUT/gpu_tests.log
timeout 280 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 120 python tools/bf16_vs_f16.py > $OUT/bf16_vs_f16.log 2>&1; tail -12 $OUT/bf16_vs_f16.log
timeout 120 python tools/launch_floor.py > $OUT/launch_floor.log 2>&1; tail -8 $OUT/launch_floor.log
bash tools/session_r05_prof.sh ${TAG}_prof > $OUT/prof_session.log 2>&1; tail -30 $OUT/prof_session.log | cut -c1-240
