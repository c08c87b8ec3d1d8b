#!/bin/bash
# round 3: backward kernel experiments (timing by ablation, loop counters, kernel stats)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py -q -m gpu --maxfail=5 --timeout=400 -x > $OUT/pytest_c.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_c.log
for ab in 0 1 2 4; do
  GRPG_BWD_ABLATE=$ab timeout 120 python tools/bench_train.py --steps 12 --warmup 3 > $OUT/train_ablate$ab.json 2>&1; echo "ablate $ab: $(tail -1 $OUT/train_ablate$ab.json | cut -c80-230)"
done
GRPG_BWD_STATS=1 timeout 120 python tools/bench_train.py --steps 2 --warmup 1 2>&1 | grep "bwd stats" | tail -2
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_c -o stats -- python $ROOT/tools/bench_train.py --steps 10 --warmup 3 > $OUT/prof_train_c.json 2> $OUT/prof_train_c.err)
python - <<'PY'
import csv,os
f=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/prof_train_c/stats_kernel_stats.csv")
try:
    rows=list(csv.DictReader(open(f)))
    for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:6]:
        print("%9.1f us x%s %s"%(float(r["AverageNs"])/1000,r["Calls"],r["Name"][:80]))
    for r in rows:
        if "publish" in r["Name"] or "frame_init" in r["Name"]: print("%9.1f us x%s %s"%(float(r["AverageNs"])/1000,r["Calls"],r["Name"][:80]))
except Exception as e: print("no stats",e)
PY
