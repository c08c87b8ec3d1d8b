#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of the bench workload.
# Outputs land in gpurun_out/prof_*; summaries worth keeping are copied into profiles/ by hand.
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
STEPS=${STEPS:-20}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- \
    python $ROOT/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery --no-secondary > $OUT/prof_stats_bench.json 2> $OUT/prof_stats.err
echo "stats rc=$?"
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES_EQ_64" \
            "GRBM_GUI_ACTIVE FETCH_SIZE" \
            "GRBM_COUNT WRITE_SIZE" \
            "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/prof_pmc$i -o pmc -- \
      python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stage-timing --no-train --no-strong --no-delivery --no-secondary > /dev/null 2> $OUT/prof_pmc$i.err
  echo "pmc$i rc=$?"
done
# config 5 (train forward + backward): kernel stats + HBM counters of the backward kernels
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_stats -o stats -- \
    python $ROOT/tools/bench_train.py --steps 10 --warmup 3 > $OUT/prof_train_bench.json 2> $OUT/prof_train_stats.err
echo "train stats rc=$?"
j=0
for ctrs in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  j=$((j+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/prof_train_pmc$j -o pmc -- \
      python $ROOT/tools/bench_train.py --steps 3 --warmup 1 > /dev/null 2> $OUT/prof_train_pmc$j.err
  echo "train pmc$j rc=$?"
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
