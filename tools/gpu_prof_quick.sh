#!/bin/bash
# Quick kernel-time profile of the bench frame loop (rocprofv3 --kernel-trace --stats).
# usage: bash tools/gpu_prof_quick.sh <tag> [env assignments...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o stats -- \
    python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery ${PROF_ARGS:-} \
    > $OUT/prof_${tag}_bench.json 2> $OUT/prof_$tag.err
echo "prof $tag rc=$?"
f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("%-70s calls=%5s avg_us=%8.2f total_ms=%8.3f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
