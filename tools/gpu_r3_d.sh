#!/bin/bash
# round 3: backward kernel -- wave lifetimes + PMC
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
GRPG_BWD_STATS=1 timeout 120 python tools/bench_train.py --steps 2 --warmup 1 2>&1 | grep "bwd stats" | tail -1
export TMPDIR=/tmp
cd /tmp
j=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE"; do
  j=$((j+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/prof_train_d_pmc$j -o pmc -- \
      python $ROOT/tools/bench_train.py --steps 3 --warmup 1 > /dev/null 2> $OUT/prof_train_d_pmc$j.err
  echo "train pmc$j rc=$?"
done
python - <<'PY'
import csv,glob,os,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/prof_train_d_pmc*/pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("grpg::","")
        if "backward" in k or "render_forward" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print("==",k)
    for c,v in sorted(agg[k].items()): print("   %-26s %16.1f (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
