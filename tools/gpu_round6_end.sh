#!/bin/bash
# round 6 evidence in one call: tools/gpu_round_end.sh (suite, smoke, driver bench x2, --steps 200, side benches,
# rocprofv3 stats + PMC passes) + the one-stream kernel statistics + the layered-frame bench and its kernel times +
# the backward's per-class reduction counters + the fill's per-phase trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT
bash tools/gpu_round_end.sh
PROF_ARGS="--streams 1 --no-secondary" bash tools/gpu_prof_quick.sh one_stream 2>&1 | tail -26
cd $ROOT
timeout 300 python tools/bench_layers.py > $OUT/bench_layers.json 2> $OUT/bench_layers.err; echo "layers rc=$?"; cut -c1-600 $OUT/bench_layers.json
bash tools/gpu_prof_layers.sh actors 2>&1 | head -4; bash tools/gpu_prof_layers.sh legacy 2>&1 | head -4
cd $ROOT
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_trace.so GRPG_BWD_STATS=1 timeout 300 python tools/bench_train.py --steps 2 --warmup 1 > /dev/null 2> $OUT/bwd_stats.txt
grep "bwd stats" $OUT/bwd_stats.txt | tail -5 > $OUT/bwd_reduction_stats.txt; cat $OUT/bwd_reduction_stats.txt
LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_filltrace.so timeout 200 python tools/fill_trace.py > $OUT/fill_trace.txt 2>&1; tail -3 $OUT/fill_trace.txt
# the host-destination frame: link micro-benchmark, the simulator leg per drain setting, the render launch's kernel time
cd $ROOT
[ -x tools/ubench/host_store ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/host_store tools/ubench/host_store.hip
timeout 120 tools/ubench/host_store > $OUT/host_store.json 2> $OUT/host_store.err; cat $OUT/host_store.json
bash tools/gpu_r6_drain.sh 0 16 2>&1 | tail -4
bash tools/gpu_prof_sim_env.sh 0 16 2>&1 | grep -E "drain_wgs|render_forward"
