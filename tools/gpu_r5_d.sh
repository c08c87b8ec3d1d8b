#!/bin/bash
# round 5: semantic planes on the matrix pipe -- parity, then the S = 15 timing; the throughput part of the render
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sweep.py -m gpu -x -q -k "semantic or fixture or sweep" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/bench_semantic.py 2>/dev/null | tee $OUT/r5d_bench_semantic.json
bash tools/gpu_ab_variants.sh "no0" 2 "--streams 1 --no-secondary" 2>&1 | tee $OUT/r5d_ab.txt
