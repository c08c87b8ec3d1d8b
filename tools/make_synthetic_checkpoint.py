"""Write a synthetic street scene as a trained-model file in the REFERENCE's layout (a .pth state dict
or a point_cloud.ply, gaussianrpg_amd/checkpoint.py), e.g. to try `bench.py --checkpoint` without a
trained Waymo scene:   python tools/make_synthetic_checkpoint.py /tmp/iteration_30000.pth [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import checkpoint as ckpt, harness as hz
from gaussianrpg_amd.composed import ModelParams

path = sys.argv[1]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
sc = hz.street_scene(P, seed=2, sh_degree=1)
op = sc.opacity.clamp(1e-6, 1 - 1e-6)
raw = ModelParams(sc.means3D, torch.log(sc.scales), sc.rotations, torch.log(op / (1 - op)),
                  sc.shs[:, :1, :].contiguous(), sc.shs[:, 1:, :].contiguous())
if path.endswith(".ply"):
    ckpt.write_ply(path, {"background": raw})
else:
    torch.save(ckpt.state_dict_of({"background": raw}), path)
print("wrote", path, "P =", P)
