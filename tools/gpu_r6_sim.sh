#!/bin/bash
# bench.py with the simulator_frame leg (short), printing the legs of interest
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} > $OUT/r6_bench.json 2> $OUT/r6_bench.err; echo "bench rc=$?"
tail -3 $OUT/r6_bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r6_bench.json")).read().strip().splitlines()[-1])
print("value %.1f ms/step %.4f" % (d["value"], d["ms_per_step"]), "single_stream", d.get("single_stream"))
print("render_all", d.get("render_all"))
print("simulator_frame", json.dumps(d.get("simulator_frame"), indent=1))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "serial sum", d["serial_stage_sum_ms"])
PY
