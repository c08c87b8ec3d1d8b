#!/bin/bash
# round 5: blend backward -- live box on the scalar unit, wave totals on the matrix pipe: parity, then A/B against the folding tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_cabi_backward.py tests/test_compose.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do
  for v in cur bwdtree; do
    pre=""; [ $v != cur ] && pre="$ROOT/build/variants/libgrpg_rasterizer_$v.so"
    LD_PRELOAD=$pre timeout 300 python tools/bench_train.py 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k or 'kernel' in k})"
  done
done 2>&1 | tee $OUT/r5e_bwd_ab.txt
