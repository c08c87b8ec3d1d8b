#!/bin/bash
# rocprofv3 average of kernels matching $1 on the one-stream bench, for this tree and each variant given
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
pat=$1; shift
for v in base "$@"; do
  pre=""; [ $v != base ] && pre="$ROOT/build/variants/libgrpg_rasterizer_$v.so"
  for rep in 1 2; do
  LD_PRELOAD=$pre timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_${v}_$rep -o stats -- python $ROOT/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-train --no-strong --no-delivery --no-secondary > /dev/null 2> $OUT/ab_$v.err
  f=$(find $OUT/ab_${v}_$rep -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" "$pat" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in r["Name"] and int(r["Calls"]) > 50:
        print("%-10s %-50s calls=%4s avg_us=%8.2f" % (sys.argv[2], r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
  done
done
