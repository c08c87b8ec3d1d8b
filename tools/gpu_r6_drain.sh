#!/bin/bash
# the host-destination frame: tests, then bench.py's simulator_frame leg per GRPG_DRAIN_WGS value (0 = no drain)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_frame.py -x -q -m gpu 2>&1 | tail -5
for n in "$@"; do
  GRPG_DRAIN_WGS=$n timeout 300 python tools/prof_sim.py one_call_rgb_host > $OUT/drain_$n.json 2> $OUT/drain_$n.err || tail -3 $OUT/drain_$n.err
  echo "drain_wgs=$n $(tail -1 $OUT/drain_$n.json)"
done
timeout 300 python tools/prof_sim.py one_call_rgb | tail -1
