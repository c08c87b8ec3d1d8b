#!/bin/bash
# round 5, call A: the whole GPU suite on the pipeline build, then A/B of the class-0 schedules on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/r5a_pytest.txt
cat $OUT/r5a_pytest.txt
bash tools/gpu_ab_variants.sh "pcpair pipe4096 pipe16384" 2 "--streams 1 --no-secondary" 2>&1 | tee $OUT/r5a_ab_serial.txt
bash tools/gpu_ab_variants.sh "pcpair pipe4096" 2 2>&1 | tee $OUT/r5a_ab_streams3.txt
