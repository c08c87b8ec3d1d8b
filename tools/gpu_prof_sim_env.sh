#!/bin/bash
# kernel times of prof_sim.py one_call_rgb_host per GRPG_DRAIN_WGS value: gpu_prof_sim_env.sh 0 16 ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for n in "$@"; do
GRPG_DRAIN_WGS=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_drain_$n -o stats -- python $ROOT/tools/prof_sim.py one_call_rgb_host > $OUT/prof_drain_$n.json 2> $OUT/prof_drain_$n.err
tail -1 $OUT/prof_drain_$n.json
f=$(find $OUT/prof_drain_$n -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$n" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if int(r["Calls"])>=20 and int(r["Calls"])<=30]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("drain_wgs", sys.argv[2], "sum of per-frame kernels %.1f us" % (sum(float(r["AverageNs"]) for r in rows)/1e3))
for r in rows[:3]:
    print("   %-70s calls=%4s avg_us=%8.2f min=%8.2f max=%8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
