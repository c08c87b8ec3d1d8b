# quick A/B of the frame loop: bash tools/gpu_ab.sh tag [ENV=val ...]   (one bench line per call)
cd $GRAFT_REPO_ROOT
tag=$1; shift
env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery ${AB_ARGS:-} > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
python - gpurun_out/ab_$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d.get("stages_ms_serial") or {}
    print(sys.argv[1].split('/')[-1], "fps=%.1f ms=%.3f" % (d["value"], d["ms_per_step"]),
          "serial:", " ".join("%s=%.3f" % (k[:6], v) for k, v in s.items() if v), "sum=%.3f" % d["serial_stage_sum_ms"], "lat", (d.get("frame_latency") or {}).get("median_ms"))
except Exception as e:
    print(sys.argv[1], "unparsable", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
