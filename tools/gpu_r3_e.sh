#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
run() { echo "== $*"; env "$@" timeout 120 python tools/bench_train.py --steps 12 --warmup 3 2>&1 | tail -1 | cut -c80-200; }
run GRPG_X=0
run GRPG_BWD_WAVES=5
run GRPG_BWD_WAVES=1
run GRPG_BWD_LIGHT=4
timeout 300 python -m pytest tests/test_gpu_backward.py -q -m gpu -x 2>&1 | tail -2
