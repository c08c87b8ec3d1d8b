#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_binning.py tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_layers.py -q -m gpu -x --timeout=600 2>&1 | tail -3
LD_PRELOAD=build/variants/libgrpg_rasterizer_filltrace.so python tools/fill_trace.py 2>&1 | tail -2
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fill -o stats -- python $ROOT/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-train --no-strong --no-delivery --no-secondary > $OUT/prof_fill_bench.json 2> $OUT/prof_fill.err
f=$(find $OUT/prof_fill -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in ("hb_fill","render_forward","preprocess_kernel","hb_count","depth_scatter")):
        print("%-64s calls=%4s avg_us=%8.2f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
