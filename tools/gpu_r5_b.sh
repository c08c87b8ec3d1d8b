#!/bin/bash
# round 5, call B: class-0 trace of the pipeline, A/B pipeline / pairs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_pc_timeout.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_trace.so GRPG_RENDER_TRACE=$OUT/trace_pl.bin timeout 300 python tools/trace_render.py 2>&1 | tail -1
python tools/trace_class0.py $OUT/trace_pl.bin 2>&1 | tee $OUT/r5b_trace_class0.txt
bash tools/gpu_ab_variants.sh "${1:-pcpair pipe4096}" 2 "--streams 1 --no-secondary" 2>&1 | tee $OUT/r5b_ab_serial.txt
