# hierarchical binning bring-up: parity tests then A/B against the sort path
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/hier_tests.txt
cat gpurun_out/hier_tests.txt
bash tools/gpu_ab.sh hier
bash tools/gpu_ab.sh sort GRPG_BINNING=sort
PROF_ARGS='--streams 1' bash tools/gpu_prof_quick.sh hier1
