#!/bin/bash
# forward parity tests + a short bench run (render stage time, frames/s): bash tools/gpu_fwd_check.sh [notest]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
if [ "${1:-}" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
fi
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery --no-secondary 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fps %.1f  ms/frame %.4f  render_serial %.4f  serial_sum %.4f  op_dev %.4f' % (j['value'], j['ms_per_step'], j['stages_ms_serial']['render'], j['serial_stage_sum_ms'], j['op_device_time']['median_ms']))
print('stages', {k: (round(v,4) if v else v) for k,v in j['stages_ms_serial'].items()})
"
done
