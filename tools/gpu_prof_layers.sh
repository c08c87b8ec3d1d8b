#!/bin/bash
# kernel times of the layered frame: bash tools/gpu_prof_layers.sh <case> [variant]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
c=$1; v=${2:-}
pre=""; [ -n "$v" ] && pre="$ROOT/build/variants/libgrpg_rasterizer_$v.so"
LD_PRELOAD=$pre timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_layers_$c$v -o stats -- python $ROOT/tools/prof_layers.py $c 20 > /dev/null 2> $OUT/prof_layers_$c$v.err
f=$(find $OUT/prof_layers_$c$v -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$c$v" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=0
for r in rows[:14]:
    print("%-10s %-64s calls=%4s avg_us=%8.2f" % (sys.argv[2], r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
