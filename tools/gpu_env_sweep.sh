#!/bin/bash
# short bench (render stage, frames/s) under a list of environment settings: bash tools/gpu_env_sweep.sh "A=1 B=2" "A=3" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for e in "$@"; do
  env $e timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery --no-secondary 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s fps %.1f  render %.4f  fill %.4f  serial %.4f' % ('$e', j['value'], j['stages_ms_serial']['render'], j['stages_ms_serial']['tile_count_fill'], j['serial_stage_sum_ms']))"
done
