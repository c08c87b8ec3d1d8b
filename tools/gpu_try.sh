cd $GRAFT_REPO_ROOT
AB_ARGS='--steps 100' bash tools/gpu_ab.sh hm256
AB_ARGS='--steps 100' bash tools/gpu_ab.sh hm128 GRPG_HEAVY_MIN=128
AB_ARGS='--steps 100' bash tools/gpu_ab.sh hm512 GRPG_HEAVY_MIN=512
AB_ARGS='--steps 100' bash tools/gpu_ab.sh pc16 GRPG_PC_MUL=16
AB_ARGS='--steps 100' bash tools/gpu_ab.sh pc64 GRPG_PC_MUL=64
AB_ARGS='--steps 100' bash tools/gpu_ab.sh classic GRPG_DEPTH_SORT=classic
