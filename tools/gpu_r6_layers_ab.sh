#!/bin/bash
# layered frame: stage times of this tree and of experiment variants (LD_PRELOAD), one JSON line each
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for v in base "$@"; do
  if [ $v = base ]; then pre=""; else pre="$ROOT/build/variants/libgrpg_rasterizer_$v.so"; fi
  LD_PRELOAD=$pre timeout 300 python tools/bench_layers.py > $OUT/r6_layers_$v.json 2> $OUT/r6_layers_$v.err
  python - "$v" "$OUT/r6_layers_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-10s fwd %.3f layers %.3f (render %.3f, pre %.3f) none %.3f (render %.3f)" % (sys.argv[1], d["forward_ms"], d["forward_layers_ms"], d["stage_ms_layers"][6], d["stage_ms_layers"][0], d["forward_layers_no_objects_ms"], d["stage_ms_layers_no_objects"][6]), {k:round(v,3) for k,v in d.items() if k.startswith("actors") and isinstance(v,float)})
PY
done
