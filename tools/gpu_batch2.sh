cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py -q -m gpu -x --timeout=300 -k "forward_matches_oracle or equal_depth or config2 or config0 or full_size or giant" > gpurun_out/pytest_b2.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_b2.log
bash tools/gpu_prof_quick.sh fat GRPG_DEPTH_SORT=fat
bash tools/gpu_prof_quick.sh classic GRPG_DEPTH_SORT=classic
