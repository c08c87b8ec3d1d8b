#!/bin/bash
# A/B on one box: this tree's library vs LD_PRELOADed variants (build/variants/libgrpg_rasterizer_<v>.so)
#   bash tools/gpu_ab_variants.sh "<v1> <v2> ..." [reps] [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT
variants=$1; reps=${2:-2}; extra=${3:-}
for rep in $(seq 1 $reps); do
for v in cur $variants; do
  pre=""; [ $v != cur ] && pre="$ROOT/build/variants/libgrpg_rasterizer_$v.so"
  LD_PRELOAD=$pre timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery $extra > $OUT/abv_${v}_$rep.json 2> $OUT/abv_${v}_$rep.err
  python - $OUT/abv_${v}_$rep.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-8s" % sys.argv[2], "fwd %.1f def %.1f" % (d["entry_points"]["forward"], d["entry_points"]["forward_deferred"] or 0), "serial", {k[:6]: round(v,4) for k,v in d["stages_ms_serial"].items() if v}, "sum %.4f" % d["serial_stage_sum_ms"], "opdev %.4f" % d["op_device_time"]["median_ms"])
except Exception as e: print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
done
done
