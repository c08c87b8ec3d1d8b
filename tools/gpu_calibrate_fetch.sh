#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on this build's access patterns (two separate PMC passes, --kernel-trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B=$ROOT/tools/ubench/gather_fetch
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -o $B $ROOT/tools/ubench/gather_fetch.hip
timeout 120 $B 22 > $OUT/calib_true.json
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o pmc -- $B 22 > /dev/null 2> $OUT/calib_fetch.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o pmc -- $B 22 > /dev/null 2> $OUT/calib_write.err
python3 $ROOT/tools/calibrate_fetch.py $OUT/calib_true.json $OUT/calib_fetch $OUT/calib_write $OUT/calib_fetch.json
