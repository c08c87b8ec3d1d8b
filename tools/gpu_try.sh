cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh rs2 GRPG_RENDER_STREAM=2
AB_ARGS='--steps 200' bash tools/gpu_ab.sh rs2_200 GRPG_RENDER_STREAM=2
AB_ARGS='--streams 2' bash tools/gpu_ab.sh rs2_s2 GRPG_RENDER_STREAM=2
AB_ARGS='--streams 2' bash tools/gpu_ab.sh base_s2
python - <<'PY'
import json
for t in ("rs2","rs2_200","rs2_s2","base_s2"):
    d=json.loads(open("gpurun_out/ab_%s.json"%t).read().strip().splitlines()[-1]); print(t, round(d["value"],1), "roof", round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],3), "frame", round(d["frame_roofline"]["frac"],3), "lat", round(d["frame_latency"]["median_ms"],3), "sum", round(d["serial_stage_sum_ms"],3))
PY
