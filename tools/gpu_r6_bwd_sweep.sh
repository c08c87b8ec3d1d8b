#!/bin/bash
# backward sweep with fresh draws, serial, failures listed:  gpu_r6_bwd_sweep.sh <offset> <draws> [variant]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
export GRPG_SWEEP_OFFSET=${1:-71000} GRPG_SWEEP_BACKWARD=${2:-40} GRPG_SWEEP_FORWARD=1
v=${3:-cur}
pre=""; [ $v != cur ] && pre="$ROOT/build/variants/libgrpg_rasterizer_$v.so"
LD_PRELOAD=$pre timeout ${4:-800} python -m pytest tests/test_gpu_sweep.py -k backward_sweep -q -m gpu --timeout=300 --tb=line -rf --durations=5 2>&1 | tail -60 > $OUT/bwd_sweep_$v.log
echo "== $v"; grep -E "passed|failed|Error|error" $OUT/bwd_sweep_$v.log | tail -12 | cut -c1-600
