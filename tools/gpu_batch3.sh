cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py -q -m gpu -x --timeout=300 -k "forward_matches_oracle or equal_depth or config2 or config0 or full_size or giant or alternative" > gpurun_out/pytest_b3.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_b3.log
bash tools/gpu_prof_quick.sh fat2 GRPG_DEPTH_SORT=fat
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery"
for v in fat classic; do
  GRPG_DEPTH_SORT=$v timeout 300 $B > gpurun_out/b3_$v.json 2> gpurun_out/b3_$v.err
done
GRPG_RADIX_FIRST_BITS=6 timeout 300 $B > gpurun_out/b3_bits6.json 2> gpurun_out/b3_bits6.err
GRPG_RADIX_FIRST_BITS=8 timeout 300 $B > gpurun_out/b3_bits8.json 2> gpurun_out/b3_bits8.err
for f in gpurun_out/b3_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d.get("stages_ms_serial") or {}
    print(sys.argv[1].split('/')[-1], "fps=%.1f ms=%.3f" % (d["value"], d["ms_per_step"]),
          "serial:", " ".join("%s=%.3f" % (k[:6], v) for k, v in s.items() if v), "sum=%.3f" % d["serial_stage_sum_ms"], "lat", (d.get("frame_latency") or {}).get("median_ms"))
except Exception as e:
    print(sys.argv[1], "unparsable", e)
PY
done
