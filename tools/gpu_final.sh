# Round-end evidence: GPU suite, smoke, the driver's invocation three times, the long run, profiles.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --timeout=300 > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_driver_$i.json 2> gpurun_out/final_driver_$i.err; echo "driver run $i rc=$?"
done
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > gpurun_out/final_200.json 2> gpurun_out/final_200.err; echo "200 rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --deferred 0 --no-cpu-baseline --no-train --no-strong --no-delivery > gpurun_out/final_streams1.json 2> gpurun_out/final_streams1.err
timeout 300 python bench.py --steps 20 --warmup 5 --streams 2 --deferred 0 --no-cpu-baseline --no-train --no-delivery > gpurun_out/final_sync2.json 2> gpurun_out/final_sync2.err
timeout 300 python bench.py --steps 20 --warmup 5 --binning-mode 1 --deferred 0 --streams 2 --no-cpu-baseline --no-train --no-strong --no-delivery > gpurun_out/final_exact.json 2> gpurun_out/final_exact.err
bash tools/profile_gpu.sh > gpurun_out/profile_run.log 2>&1; grep "rc=" gpurun_out/profile_run.log
for f in gpurun_out/final_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "fps=%.1f ms=%.3f" % (d["value"], d["ms_per_step"]), "sum=%.3f" % (d.get("serial_stage_sum_ms") or 0),
          "lat", (d.get("frame_latency") or {}).get("median_ms"), "strong", (d.get("strong_scaling") or {}).get("frames_per_s"),
          "train", {k: round(v, 3) for k, v in (d.get("train") or {}).items() if k.endswith("median")}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unparsable", e)
PY
done
# the N > 1 flow of the driver's launch line, two ranks on this box's one GPU (gloo stands in for RCCL)
GRPG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_2rank_gloo.json 2> gpurun_out/final_2rank_gloo.err; echo "2-rank rc=$?"; tail -c 600 gpurun_out/final_2rank_gloo.json
