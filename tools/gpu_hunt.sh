#!/bin/bash
# a bug hunt with fresh draws of the randomised sweeps (tests/test_gpu_sweep.py, tests/test_gpu_layers.py)
cd ${GRAFT_REPO_ROOT:-.}
export GRPG_SWEEP_OFFSET=${1:-40000} GRPG_SWEEP_FORWARD=${2:-240} GRPG_SWEEP_BACKWARD=${3:-60} GRPG_SWEEP_LAYERS=${4:-300}
# serial by default (GRPG_HUNT_WORKERS=4 for xdist): every worker's float64 autograd takes all of the host's threads, and on
# a box with few idle cores four workers ran 25x slower than one (round 6: two hunts cut by their time limits)
W=${GRPG_HUNT_WORKERS:-0}; NW=""; [ "$W" != "0" ] && NW="-n $W"
timeout 2400 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_layers.py -q -m gpu --maxfail=15 --timeout=600 --tb=line -rf $NW 2>&1 | tail -25
