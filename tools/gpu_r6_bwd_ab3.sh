#!/bin/bash
# kernel times of the train step for this tree's library and variants, alternating on one box: gpu_r6_bwd_ab3.sh "<variants>" [reps]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
for rep in $(seq 1 ${2:-2}); do
  for v in cur $1; do
    pre="GRPG_DUMMY=1"; [ $v != cur ] && pre="LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_$v.so"
    bash tools/gpu_prof_train.sh ${v}_$rep $pre | grep render_backward
  done
done
