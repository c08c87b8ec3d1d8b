#!/bin/bash
# round 5 evidence in one call: tools/gpu_round_end.sh + the one-stream kernel statistics the verdict asks for
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT
bash tools/gpu_round_end.sh
PROF_ARGS="--streams 1 --no-secondary" bash tools/gpu_prof_quick.sh one_stream 2>&1 | tail -26
