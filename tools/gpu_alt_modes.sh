#!/bin/bash
# the GPU suite under the non-default policies: sort-based binning, exact (wait-then-carve) capacity mode
cd ${GRAFT_REPO_ROOT:-.}
if [ "${1:-all}" != "exact" ]; then
echo "== GRPG_BINNING=sort"; GRPG_BINNING=sort timeout 1200 python -m pytest tests -q -m gpu --maxfail=20 --timeout=500 -k "not psnr and not soak" 2>&1 | tail -6
fi
echo "== GRPG_SYNC_R=1"; GRPG_SYNC_R=1 timeout 1200 python -m pytest tests -q -m gpu --maxfail=20 --timeout=500 -k "not psnr and not soak" 2>&1 | tail -12
