cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t2 -o stats -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 10 --warmup 3 2>/dev/null | tail -1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_t2 -name "*kernel_stats.csv" | head -1)
grep -E "preprocess_backward|render_backward" $f | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "not psnr" 2>&1 | tail -2
