#!/bin/bash
# everything the round's committed numbers come from, in one gpurun call:
# full GPU suite + smoke + the driver's bench invocation x2 + --steps 200 (tools/gpu_final.sh), the
# composition / train / semantic side benches, rocprofv3 stats + PMC passes (tools/profile_gpu.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
rm -f $OUT/parity_stats_*.json
bash tools/gpu_final.sh
timeout 300 python tools/bench_compose.py > $OUT/bench_compose.json 2> $OUT/bench_compose.err; echo "compose rc=$?"; tail -c 900 $OUT/bench_compose.json
timeout 200 python tools/bench_train.py --steps 20 --warmup 3 > $OUT/bench_train.json 2>/dev/null; cat $OUT/bench_train.json
timeout 200 python tools/bench_semantic.py > $OUT/bench_semantic.json 2>/dev/null; cat $OUT/bench_semantic.json
bash tools/profile_gpu.sh 2>&1 | grep "rc="
