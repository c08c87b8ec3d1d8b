#!/bin/bash
# A/B on one box: this tree's library vs the round-4 library (build/variants/libgrpg_rasterizer_r4.so, built from
# commit b97ad69's csrc/ and LD_PRELOADed under this tree's binding): serial stages, single-frame device time, the
# three-stream loop; then the class-0 chain alone and the semantic bench with both
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT
bash tools/gpu_ab_variants.sh "r4" 3 2>&1 | tee $OUT/r5_ab_r4.txt
bash tools/gpu_ab_variants.sh "r4" 2 "--streams 1 --no-secondary" 2>&1 | tee $OUT/r5_ab_r4_serial.txt
