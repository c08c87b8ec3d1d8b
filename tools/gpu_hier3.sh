cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/gpu_ab.sh hier
AB_ARGS='--steps 200' bash tools/gpu_ab.sh h_200
