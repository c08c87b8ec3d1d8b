#!/bin/bash
# round 6: the blend backward with ONE scalar accumulator per pixel against the per-channel form (variant bwdold)
#   backward tests first, then rocprof kernel times of the train step, alternating on one box; then a hunt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_cabi_backward.py tests/test_gpu_smoke_script.py tests/test_compose.py tests/test_gpu_configs.py -m gpu -q --maxfail=10 --timeout=900 2>&1 | tail -30 > $OUT/bwd_ab_tests.log
tail -5 $OUT/bwd_ab_tests.log
cp $(ls -t $OUT/parity_stats_*.json | head -1) $OUT/bwd_ab_parity_stats.json
for rep in 1 2; do
  for v in cur bwdold; do
    [ $v != cur ] && [ ! -f $ROOT/build/variants/libgrpg_rasterizer_$v.so ] && continue
    pre=""; [ $v != cur ] && pre="LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_$v.so"
    bash tools/gpu_prof_train.sh ${v}_$rep $pre GRPG_DUMMY=1
  done
done
bash tools/gpu_hunt.sh ${1:-71000} ${2:-20} ${3:-150} ${4:-10} > $OUT/bwd_ab_hunt.log 2>&1; tail -4 $OUT/bwd_ab_hunt.log
