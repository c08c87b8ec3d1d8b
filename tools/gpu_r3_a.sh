#!/bin/bash
# round 3, first GPU call: whole GPU suite, VALU-rate ubench, one default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu --maxfail=10 --timeout=400 --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
timeout 120 tools/ubench/valu_rate $OUT/valu_rate.json > $OUT/valu_rate.txt 2>&1; echo "ubench rc=$?"
cat $OUT/valu_rate.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_a.json 2> $OUT/bench_a.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench_a.json
