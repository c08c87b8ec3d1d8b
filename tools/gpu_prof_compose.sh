#!/bin/bash
# kernel times of the composition bench (rocprofv3 --kernel-trace --stats): bash tools/gpu_prof_compose.sh <tag> [ENV=val ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; shift
export TMPDIR=/tmp
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profc_$tag -o stats -- \
    python $ROOT/tools/bench_compose.py > $OUT/profc_$tag.json 2> $OUT/profc_$tag.err
python3 - "$OUT/profc_$tag/stats_kernel_stats.csv" "$tag" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "backward" in r["Name"] or "render_forward" in r["Name"] or "preprocess" in r["Name"]:
        print("%-8s %-70s calls=%4s avg_us=%8.2f" % (sys.argv[2], r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
python3 -c "
import json,sys
j=json.loads(open('$OUT/profc_$tag.json').read().strip().splitlines()[-1])
print('$tag', 'eval fused %.3f  train fused %.3f' % (j['fused_composed_op_ms']['median_ms'], j['train_step']['fused_composed_op_ms']['median_ms']))
"
