#!/bin/bash
# everything the round's committed numbers come from, in one gpurun call:
# rocprofv3 stats + PMC passes (tools/profile_gpu.sh), full GPU suite + smoke + the driver's bench
# invocation x2 + --steps 200 (tools/gpu_final.sh), the composition bench, the VALU ubench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
bash tools/gpu_final.sh
timeout 300 python tools/bench_compose.py > $OUT/bench_compose.json 2> $OUT/bench_compose.err; echo "compose rc=$?"; tail -c 900 $OUT/bench_compose.json
timeout 200 python tools/bench_train.py --steps 20 --warmup 3 > $OUT/bench_train.json 2>/dev/null; cat $OUT/bench_train.json
bash tools/profile_gpu.sh 2>&1 | grep "rc="
