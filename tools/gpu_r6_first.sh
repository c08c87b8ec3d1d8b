#!/bin/bash
# round 6, first GPU call: the new true-size parity tests, the layered-frame baseline, the backward's per-class
# reduction counters (trace build) and the FETCH_SIZE calibration
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_smoke_script.py tests/test_gpu_layers.py tests/test_bench_self_launch.py -q -m gpu --timeout=600 > $OUT/r6_first_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/r6_first_pytest.log
timeout 300 python tools/bench_layers.py > $OUT/r6_layers_base.json 2> $OUT/r6_layers_base.err; echo "layers rc=$?"; cat $OUT/r6_layers_base.json
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_trace.so GRPG_BWD_STATS=1 timeout 300 python tools/bench_train.py --steps 2 --warmup 1 > $OUT/r6_bwd_stats.json 2> $OUT/r6_bwd_stats.txt
echo "bwd stats rc=$?"; grep "bwd stats" $OUT/r6_bwd_stats.txt | tail -5
bash tools/gpu_calibrate_fetch.sh 2>&1 | tail -80
