#!/bin/bash
# smoke + the driver's bench invocation once: gpu_bench_once.sh <n>  -> gpurun_out/bench_final_<n>.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; n=${1:-3}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final_$n.json 2> gpurun_out/bench_final_$n.err; echo rc=$?
python - $n <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_final_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f ms %.4f roofline %.3f traffic %.3e valu %.3f | train bwd roofline %.3f render_backward %.4f ms | cpu %.4f" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_valu"]["frac"],
    d["train"]["backward_roofline"]["frac"], d["train"]["backward_kernels_ms"]["render_backward_kernel"], d["cpu_baseline"]["value"]))
PY
