cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_binning.py tests/test_compose.py tests/test_gpu_backward.py -x -q -m gpu -k "not psnr" 2>&1 | tail -3
bash tools/gpu_ab.sh hier
AB_ARGS='--steps 200' bash tools/gpu_ab.sh h_200
