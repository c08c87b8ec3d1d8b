#!/bin/bash
# what do the two HIP events around the render launch cost the timed region?  bench.py with / without --no-stage-timing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2 3; do
  for f in "" "--no-stage-timing"; do
    timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery --no-secondary $f 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-18s value %.1f  ms %.4f' % ('$f' or 'events', d['value'], d['ms_per_step']))"
  done
done
