#!/bin/bash
# a bug hunt with fresh draws of the randomised sweeps (tests/test_gpu_sweep.py, tests/test_gpu_layers.py)
cd ${GRAFT_REPO_ROOT:-.}
export GRPG_SWEEP_OFFSET=${1:-40000} GRPG_SWEEP_FORWARD=${2:-240} GRPG_SWEEP_BACKWARD=${3:-60} GRPG_SWEEP_LAYERS=${4:-300}
timeout 2400 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_layers.py -q -m gpu --maxfail=15 --timeout=600 -n 4 2>&1 | tail -25
