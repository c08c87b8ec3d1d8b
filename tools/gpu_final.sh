#!/bin/bash
# full GPU suite + smoke + the driver's bench invocation (x2) in one call
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -q -m gpu --maxfail=10 --timeout=500 --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_final_$i.json 2> $OUT/bench_final_$i.err; echo "bench$i rc=$?"
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > $OUT/bench_200.json 2> $OUT/bench_200.err; echo "b200 rc=$?"
python - <<'PY'
import json,os
root=os.environ.get("GRAFT_REPO_ROOT",".")
for n in ("bench_final_1","bench_final_2","bench_200"):
    try:
        d=json.loads(open(os.path.join(root,"gpurun_out",n+".json")).read().strip().splitlines()[-1])
        print(n,"value %.1f ms %.4f"%(d["value"],d["ms_per_step"]),"entry",{k:(round(v,1) if isinstance(v,float) else v) for k,v in d["entry_points"].items() if k in("forward","forward_deferred")},
              "roof %.3f (%.4f ms)"%(d["roofline"]["frac"],d["roofline"]["avg_launch_ms"]),"valu",(d["roofline_valu"] or {}).get("frac"),"lat",d["frame_latency"]["median_ms"] if d["frame_latency"] else None,
              "opdev",d["op_device_time"]["median_ms"] if d["op_device_time"] else None,"sum",d["serial_stage_sum_ms"])
        if d.get("train"): t=d["train"]; print("   train fwd %.3f bwd %.3f lossbwd %.3f opbwd_dev %.3f roof %.3f | op only: fwd %.3f bwd %.3f"%(t["forward_ms_median"],t["backward_ms_median"],t["loss_backward_ms_median"],t["op_backward_device_ms_median"],t["backward_roofline"]["frac"],t["op_only"]["forward_device_ms_median"],t["op_only"]["backward_device_ms_median"]), "kernels", {k:(round(v,4) if isinstance(v,float) else v) for k,v in t["backward_kernels_ms"].items() if k!="measured"})
        if d.get("cpu_baseline"): print("   cpu",d["cpu_baseline"]["value"],d["cpu_baseline"]["c_restatement"]["openmp"]["value"])
        print("   stages",{k:round(v,4) for k,v in d["stages_ms_serial"].items() if v})
    except Exception as e: print(n,"parse error",e)
PY
