#!/bin/bash
# kernel times of the train step (rocprofv3 --kernel-trace --stats) for one variant:
#   bash tools/gpu_prof_train.sh <tag> [ENV=val ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proft_$tag -o stats -- \
    python $ROOT/tools/bench_train.py --steps 20 --warmup 3 > $OUT/proft_$tag.json 2> $OUT/proft_$tag.err
f=$(find $OUT/proft_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$tag" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "backward" in r["Name"] or "render_forward" in r["Name"]:
        print("%-12s %-60s calls=%4s avg_us=%8.2f" % (sys.argv[2], r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
