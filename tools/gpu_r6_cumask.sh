#!/bin/bash
# the headline loop on CU-masked streams, one configuration per process, alternating with the plain three streams
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for c in plain3 xcd_halves_x1 xcd_halves_x2 plain3 xcd_quarters_x1 xcd_quarters_x2 cu_halves_x2 plain3 threequarters+quarter; do
  timeout 200 python tools/experiments/cu_mask_streams.py 20 $c 2>/dev/null | tail -1
done
