#!/bin/bash
# HBM counters of the render launch of the simulator frame (separate --pmc passes, --kernel-trace only):
# device bytes + copy (one_call_rgb) against the drained host frame (one_call_rgb_host)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for v in one_call_rgb one_call_rgb_host; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_sim_${v}_$tag -o pmc -- python $ROOT/tools/prof_sim.py $v > /dev/null 2> $OUT/pmc_sim_${v}_$tag.err
    f=$(find $OUT/pmc_sim_${v}_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "render_forward_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print("%-20s %-12s mean per launch %14.1f  (n=%d)" % (sys.argv[2], k, sum(v)/len(v), len(v)))
PY
  done
done
