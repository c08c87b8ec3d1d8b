#!/bin/bash
# kernel stats of bench.py's train leg alone (the reference caller's pattern) next to tools/bench_train.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trainleg -o stats -- \
  python -c "
import sys, json; sys.path.insert(0, '$ROOT')
import torch, bench
r = bench.train_leg(torch.device('cuda:0'))
print(json.dumps({k: r[k] for k in ('forward_ms_median','backward_ms_median','loss_backward_ms_median','op_backward_device_ms_median')}))
" 2> $OUT/prof_trainleg.err | tail -1
python3 - "$OUT/prof_trainleg/stats_kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%8.1f us x %4s  %s" % (float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:90]))
PY
