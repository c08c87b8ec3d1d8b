#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for v in "$@"; do
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/prof_sim_$v -o stats -- python $ROOT/tools/prof_sim.py $v > $OUT/prof_sim_$v.json 2> $OUT/prof_sim_$v.err
cat $OUT/prof_sim_$v.json | tail -1
f=$(find $OUT/prof_sim_$v -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$v" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if int(r["Calls"])>=20 and int(r["Calls"])<=30]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["AverageNs"]) for r in rows)
print(sys.argv[2], "sum of per-frame kernels %.1f us" % (tot/1e3))
for r in rows[:9]:
    print("   %-70s calls=%4s avg_us=%8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
