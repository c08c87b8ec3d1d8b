#!/bin/bash
# round 6: the pairs' consumer with two geometry blocks taking turns (no register moves per quad) against the library
# before it (variant fwdold): frame hashes of both, forward parity files, then bench A/B alternating on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; cd $ROOT; mkdir -p $OUT
python tools/frame_hash.py > $OUT/hash_cur.txt 2>/dev/null; LD_PRELOAD=$ROOT/build/variants/libgrpg_rasterizer_fwdold.so python tools/frame_hash.py > $OUT/hash_old.txt 2>/dev/null
if cmp -s $OUT/hash_cur.txt $OUT/hash_old.txt; then echo "frame hashes identical ($(wc -l < $OUT/hash_cur.txt) frames)"; else echo "FRAME HASHES DIFFER"; diff $OUT/hash_cur.txt $OUT/hash_old.txt | head; fi
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_layers.py tests/test_gpu_frame.py tests/test_gpu_smoke_script.py tests/test_gpu_pc_timeout.py -m gpu -q --maxfail=10 --timeout=600 2>&1 | tail -3
bash tools/gpu_ab_variants.sh "fwdold" 3 "--no-secondary"
