#!/bin/bash
# layered frame: parity (bit-identical to three op calls) + stage times
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_compose.py ${EXTRA_TESTS:-} -q -m gpu -x --timeout=600 > $OUT/r6_layers_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/r6_layers_pytest.log
timeout 300 python tools/bench_layers.py > $OUT/r6_layers.json 2> $OUT/r6_layers.err; echo "layers rc=$?"; cat $OUT/r6_layers.json
