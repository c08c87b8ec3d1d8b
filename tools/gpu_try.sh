cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_binning.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_ab.sh hier
PROF_ARGS='--streams 1' bash tools/gpu_prof_quick.sh h0 | grep -E 'hb_tile_scan|hb_count'
