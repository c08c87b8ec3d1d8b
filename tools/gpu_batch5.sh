cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_compose.py tests/test_gpu_forward.py -q -m gpu -x --timeout=300 -k "compose or fused or composed or forward_matches_oracle" > gpurun_out/pytest_b5.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_b5.log
timeout 300 python tools/bench_compose.py > gpurun_out/bench_compose.json 2> gpurun_out/bench_compose.err; echo "compose rc=$?"; cat gpurun_out/bench_compose.json; tail -3 gpurun_out/bench_compose.err
