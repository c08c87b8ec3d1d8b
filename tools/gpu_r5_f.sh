#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_layers.py -m gpu -x -q 2>&1 | tail -25
python - <<'PY'
import json,glob,os
fs=sorted(glob.glob("gpurun_out/parity_stats_*.json"), key=os.path.getmtime)
for r in json.load(open(fs[-1])):
    if r.get("plane")=="render_all_ms": print(r)
PY
