#!/bin/bash
# build a variant (extra hipcc flags) and profile the train step on the GPU:
#   bash tools/variant_run.sh <tag> "<flags>" [test] [ENV=val ...]
tag=$1; flags=$2; shift 2
pre=""
if [ "${1:-}" = "test" ]; then pre="timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -2;"; shift; fi
cd /root/repo
GRPG_EXTRA_HIPCC_FLAGS="$flags" python -m gaussianrpg_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
/usr/local/graft/bin/gpurun --timeout 900 -- "$pre bash tools/gpu_prof_train.sh $tag $*; cat gpurun_out/proft_$tag.json" 2>&1 | grep -E "^$tag|op_backward|GPU-minutes|passed|failed"
