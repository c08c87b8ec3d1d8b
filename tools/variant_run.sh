#!/bin/bash
# build a variant and profile BOTH training workloads (config 5 op, composed 2 M scene):
#   bash tools/variant_run2.sh <tag> "<flags>"
tag=$1; flags=$2; shift 2
cd /root/repo
GRPG_EXTRA_HIPCC_FLAGS="$flags" python -m gaussianrpg_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
/usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/gpu_prof_train.sh $tag $*; bash tools/gpu_prof_compose.sh $tag $*" 2>&1 | grep -E "^$tag.*(render_backward|render_forward_kernel<true|eval fused)|GPU-minutes"
