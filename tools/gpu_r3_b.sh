#!/bin/bash
# round 3, second GPU call: backward rewrite + early count publish
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_deferred.py tests/test_gpu_forward.py tests/test_gpu_binning.py tests/test_sky.py tests/test_compose.py -q -m gpu --maxfail=10 --timeout=400 -x > $OUT/pytest_b.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_b.log
timeout 120 tools/ubench/valu_rate $OUT/valu_rate.json > $OUT/valu_rate.txt 2>&1; echo "ubench rc=$?"
cat $OUT/valu_rate.txt
for ab in 0 1 2 4; do
  GRPG_BWD_ABLATE=$ab timeout 120 python tools/bench_train.py --steps 12 --warmup 3 > $OUT/train_ablate$ab.json 2>&1; echo "ablate $ab: $(tail -1 $OUT/train_ablate$ab.json | cut -c1-400)"
done
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_b -o stats -- python $ROOT/tools/bench_train.py --steps 10 --warmup 3 > $OUT/prof_train_b.json 2> $OUT/prof_train_b.err)
python - <<'PY'
import csv,os
f=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/prof_train_b/stats_kernel_stats.csv")
try:
    rows=list(csv.DictReader(open(f)))
    for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:8]:
        print("%9.1f us x%s %s"%(float(r["AverageNs"])/1000,r["Calls"],r["Name"][:80]))
except Exception as e: print("no stats",e)
PY
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_b.json 2> $OUT/bench_b.err; echo "bench rc=$?"
python - <<'PY'
import json,os
try:
    d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/bench_b.json")).read().strip().splitlines()[-1])
    print("value",d["value"],"ms",d["ms_per_step"]); print("entry",d["entry_points"]); print("roof",d["roofline"]["frac"],d["roofline"]["avg_launch_ms"]); print("overl",d["roofline_overlapped"]["frac"] if d["roofline_overlapped"] else None)
    print("valu",d["roofline_valu"]); print("lat",d["frame_latency"]); print("opdev",d["op_device_time"]); print("stages",d["stages_ms_serial"]); print("strong",d["strong_scaling"]); t=d["train"]; print("train",{k:t[k] for k in t if k.endswith("median") or k=="backward_roofline"})
except Exception as e: print("bench parse",e); print(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/bench_b.err")).read()[-2000:])
PY
