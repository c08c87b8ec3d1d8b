#!/bin/bash
# A/B on one box: this tree's library vs the round-3 library (build/variants/libgrpg_rasterizer_r3.so, LD_PRELOAD)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT
for rep in 1 2; do
for v in cur r3; do
  pre=""; [ $v = r3 ] && pre="$ROOT/build/variants/libgrpg_rasterizer_r3.so"
  for st in 20 200; do
    wu=5; [ $st = 200 ] && wu=20
    LD_PRELOAD=$pre timeout 300 python bench.py --steps $st --warmup $wu --no-cpu-baseline --no-train --no-strong --no-delivery > $OUT/ab_${v}_${st}_$rep.json 2> $OUT/ab_${v}_${st}_$rep.err
    python - $OUT/ab_${v}_${st}_$rep.json $v $st <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], "fwd %.1f def %.1f" % (d["entry_points"]["forward"], d["entry_points"]["forward_deferred"] or 0), "serial", {k[:6]: round(v,4) for k,v in d["stages_ms_serial"].items() if v}, "sum %.4f" % d["serial_stage_sum_ms"], "lat %.4f" % d["frame_latency"]["median_ms"], "opdev %.4f" % d["op_device_time"]["median_ms"])
except Exception as e: print(sys.argv[2], sys.argv[3], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
  done
done
done
