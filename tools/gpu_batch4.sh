cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py -q -m gpu -x --timeout=300 -k "forward_matches_oracle or equal_depth or config2 or config0 or full_size or giant" > gpurun_out/pytest_b4.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_b4.log
GRPG_RS_WIDE=3 timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py -q -m gpu -x --timeout=300 -k "forward_matches_oracle or config2 or full_size" > gpurun_out/pytest_b4w.log 2>&1; echo "pytest wide rc=$?"; tail -3 gpurun_out/pytest_b4w.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery"
GRPG_DEPTH_SORT=fat timeout 300 $B > gpurun_out/b4_fat.json 2> gpurun_out/b4_fat.err
GRPG_DEPTH_SORT=classic timeout 300 $B > gpurun_out/b4_classic.json 2> gpurun_out/b4_classic.err
for w in 1 2 3; do GRPG_RS_WIDE=$w timeout 300 $B > gpurun_out/b4_wide$w.json 2> gpurun_out/b4_wide$w.err; done
GRPG_DEPTH_SORT=fat timeout 300 $B --streams 1 > gpurun_out/b4_fat_s1.json 2> gpurun_out/b4_fat_s1.err
for f in gpurun_out/b4_*.json; do python - "$f" <<'PY'
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
