#!/bin/bash
# Per-kernel instruction counters of the frame loop: bash tools/gpu_pmc_quick.sh <tag> "<counters>" [kernel-name-regex]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; ctrs=$2; pat=${3:-.}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$tag -o pmc -- \
    python $ROOT/bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-stage-timing --no-train --no-strong --no-delivery > /dev/null 2> $OUT/pmc_$tag.err
echo "pmc $tag rc=$?"
f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv,sys,re,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:60]
    if not re.search(sys.argv[2],k): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    print(k, " ".join("%s=%.3gM" % (c, v/max(1,n[(k,c)])/1e6) for c,v in d.items()))
PY
