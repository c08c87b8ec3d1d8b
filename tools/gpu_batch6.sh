cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --maxfail=6 --timeout=300 > gpurun_out/pytest_b6.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_b6.log
