#!/bin/bash
# GPU call: the whole GPU suite, a single-stream rocprofv3 kernel profile, the driver's bench invocation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 --timeout=400 ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -${TAIL:-30} $OUT/pytest_gpu.log
fi
PROF_ARGS="--streams 1" bash tools/gpu_prof_quick.sh s1 2>&1 | cut -c1-150
cd $ROOT
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > $OUT/bench_a.json 2> $OUT/bench_a.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/bench_a.json").read().strip().splitlines()[-1])
    print("fps=%.1f ms=%.4f" % (d["value"], d["ms_per_step"]), "serial", {k: round(v,4) for k,v in (d.get("stages_ms_serial") or {}).items() if v}, "sum", d.get("serial_stage_sum_ms"), "lat", (d.get("frame_latency") or {}).get("median_ms"), "entry", d.get("entry_points",{}).get("forward_deferred"))
except Exception as e:
    print("bench unparsable", e); print(open("gpurun_out/bench_a.err").read()[-2000:])
PY
