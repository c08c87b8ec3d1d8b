#!/bin/bash
# VALU utilisation of the train step's kernels (two PMC passes, --kernel-trace only):
#   bash tools/gpu_pmc_train.sh <tag> [ENV=val ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
tag=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE"; do
  i=$((i+1))
  env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmct_${tag}_$i -o pmc -- \
      python $ROOT/tools/bench_train.py --steps 3 --warmup 1 > /dev/null 2> $OUT/pmct_${tag}_$i.err
done
python3 - "$OUT" "$tag" <<'PY'
import csv, sys, glob, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("%s/pmct_%s_*/**/*counter_collection.csv" % (out, tag), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "render_backward" in k or "render_forward" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    a = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
    gui = a.get("GRBM_GUI_ACTIVE", 0)
    line = "%s %s:" % (tag, k)
    for c in sorted(a): line += " %s=%.4g" % (c, a[c])
    if gui and "SQ_ACTIVE_INST_VALU" in a:
        line += " | valu_busy=%.3f instr_per_cycle_per_simd=%.3f" % (a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (gui / 8.0) if False else (a["SQ_ACTIVE_INST_VALU"] * 4 / 1024) / (gui / 8.0), a["SQ_INSTS_VALU"] / 1024 / (gui / 8.0))
    print(line)
PY
