#!/bin/bash
# round 6, after the backward's scalar accumulator: the round-end evidence again (tools/gpu_round_end.sh) + the
# backward's per-class counters from the trace build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT
bash tools/gpu_round_end.sh
cd $ROOT
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_trace.so GRPG_BWD_STATS=1 timeout 300 python tools/bench_train.py --steps 2 --warmup 1 > /dev/null 2> $OUT/bwd_stats.txt
grep "bwd stats" $OUT/bwd_stats.txt | tail -5 > $OUT/bwd_reduction_stats.txt; cat $OUT/bwd_reduction_stats.txt | cut -c1-400
bash tools/gpu_pmc_train.sh final 2>&1 | tail -3 | cut -c1-700
