#!/bin/bash
# One gpurun call: GPU test suite + a set of bench variants; everything lands in gpurun_out/.
# usage (from the repo root on the GPU box): bash tools/gpu_batch.sh [tests|bench|all]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
what=${1:-all}
if [ "$what" = tests ] || [ "$what" = all ]; then
  timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 --timeout=300 > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  B="python bench.py --steps 20 --warmup 5"
  timeout 600 $B > $OUT/bench_default_1.json 2> $OUT/bench_default_1.err; echo "bench1 rc=$?"
  timeout 300 $B --no-cpu-baseline --no-train > $OUT/bench_default_2.json 2> $OUT/bench_default_2.err; echo "bench2 rc=$?"
  timeout 300 $B --no-cpu-baseline --no-train > $OUT/bench_default_3.json 2> $OUT/bench_default_3.err; echo "bench3 rc=$?"
  timeout 300 $B --no-cpu-baseline --no-train --streams 1 > $OUT/bench_streams1.json 2> $OUT/bench_streams1.err; echo "s1 rc=$?"
  timeout 300 $B --no-cpu-baseline --no-train --binning-mode 1 > $OUT/bench_exact.json 2> $OUT/bench_exact.err; echo "exact rc=$?"
  GRPG_DEPTH_SORT=classic timeout 300 $B --no-cpu-baseline --no-train > $OUT/bench_classic_sort.json 2> $OUT/bench_classic_sort.err; echo "classic rc=$?"
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > $OUT/bench_200.json 2> $OUT/bench_200.err; echo "b200 rc=$?"
  for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d.get("stages_ms_serial") or {}
    print(sys.argv[1].split('/')[-1], "fps=%.1f ms=%.3f" % (d["value"], d["ms_per_step"]),
          "serial:", " ".join("%s=%.3f" % (k[:6], v) for k, v in s.items() if v), "attempts", d.get("timed_region_attempts_s"),
          "lat", (d.get("frame_latency") or {}).get("median_ms"))
except Exception as e:
    print(sys.argv[1], "unparsable", e)
PY
  done
fi
