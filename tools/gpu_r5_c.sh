#!/bin/bash
# round 5, call C: the class-0 chain alone (pipeline vs pairs), traced
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_only0trace.so GRPG_RENDER_TRACE=$OUT/trace_only0.bin timeout 300 python tools/trace_render.py 2>&1 | tail -1
python tools/trace_class0.py $OUT/trace_only0.bin 2>&1 | tee $OUT/r5c_trace_only0.txt
bash tools/gpu_ab_variants.sh "only0 only0pair" 2 "--streams 1 --no-secondary" 2>&1 | tee $OUT/r5c_ab_only0.txt
