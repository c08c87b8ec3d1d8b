#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -3
run() { echo "== $*"; env "$@" timeout 120 python tools/bench_train.py --steps 30 --warmup 4 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd %.3f bwd wall %.3f  op bwd device %.3f' % (d['forward_ms_median'], d['backward_ms_median'], d['op_backward_device_ms_median']))"; }
run GRPG_X=1
bash tools/gpu_prof_train.sh final
