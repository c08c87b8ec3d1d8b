#!/bin/bash
# round 5: forward parity on the new pairs, the class-0 chain alone (traced), A/B of the class boundary
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_pc_timeout.py tests/test_gpu_configs.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -4
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_only0trace.so GRPG_RENDER_TRACE=$OUT/trace_only0.bin timeout 300 python tools/trace_render.py 2>&1 | tail -1
python tools/trace_class0.py $OUT/trace_only0.bin 2>&1 | tee $OUT/r5c_trace_only0.txt
bash tools/gpu_ab_variants.sh "only0 pc4096 pc16384" 2 "--streams 1 --no-secondary" 2>&1 | tee $OUT/r5c_ab.txt
bash tools/gpu_ab_variants.sh "pc4096" 2 2>&1 | tee $OUT/r5c_ab3.txt
