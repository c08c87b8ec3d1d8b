"""Pure-PyTorch restatement of the reference's per-frame scene-graph composition -- TEST
INFRASTRUCTURE ONLY (checker of gaussianrpg_amd/composed.py; never imported by the product).

Follows, line by line, what ``StreetGaussianModel``'s properties compute for one frame:
  get_scaling   lib/models/street_gaussian_model.py:296-312  + gaussian_model.py:224-226 (exp)
  get_rotation  street_gaussian_model.py:314-338             + gaussian_model.py:228-230 (F.normalize),
                general_utils.py:220-238 (quaternion_raw_multiply)
  get_xyz       street_gaussian_model.py:340-365, general_utils.py:125-146 (quaternion_to_matrix)
  get_features  street_gaussian_model.py:367-383, gaussian_model.py:236-240,
                gaussian_model_actor.py:73-82 (get_features_fourier), sh_utils.py:120-130 (IDFT)
  get_opacity   street_gaussian_model.py:437-453             + gaussian_model.py:248-250 (sigmoid)
with the optional per-Gaussian flip mask of training (flip_prob) and no pose correction.  PINNED since round 4:
lib/models cannot be imported here (plyfile, roma, simple_knn, nvdiffrast missing and lib.config parses
sys.argv), but tests/golden/make_golden.py part_a_compose cuts the getters above out of their classes with
ast and runs them, in the reference's code, on stand-in objects holding a background model and two posed,
partly flipped actors -> tests/golden/ref_compose.npz; this file reproduces that output within 2e-6
(tests/test_reference_pins.py), and so does grpg_compose on the GPU (tests/test_compose.py).  Piecewise
pins from earlier rounds: IDFT (ref_idft.npz), quaternion_to_matrix and quaternion_raw_multiply
(ref_quat.npz).
"""
import torch


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def quaternion_to_matrix(r):
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), dtype=r.dtype, device=r.device)
    r_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r_ * z)
    R[:, 0, 2] = 2 * (x * z + r_ * y)
    R[:, 1, 0] = 2 * (x * y + r_ * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r_ * x)
    R[:, 2, 0] = 2 * (x * z - r_ * y)
    R[:, 2, 1] = 2 * (y * z + r_ * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def idft(time, dim):
    t = torch.tensor(float(time)).view(-1, 1).float()
    out = torch.zeros(t.shape[0], dim)
    indices = torch.arange(dim)
    even, odd = indices[::2], indices[1::2]
    out[:, even] = torch.cos(torch.pi * t * even)
    out[:, odd] = torch.sin(torch.pi * t * (odd + 1))
    return out


def compose(models, poses):
    """models: sequence with .xyz .scaling .rotation .opacity .features_dc .features_rest (CPU
    tensors); poses: None or (obj_rot[4], obj_trans[3], fourier_time) per model.
    Returns (means3D, scales, rotations, opacity, shs) as the reference's getters would."""
    xyzs, scales, rots, opas, feats = [], [], [], [], []
    for m, p in zip(models, poses):
        n = m.xyz.shape[0]
        scales.append(torch.exp(m.scaling))
        opas.append(torch.sigmoid(m.opacity.reshape(n, 1)))
        rot_local = torch.nn.functional.normalize(m.rotation)
        xyz_local = m.xyz
        flip = getattr(m, "flip", None)
        if flip is not None and p is not None:
            # street_gaussian_model.py:57-61 (flip_axis = 1, flip_matrix = quaternion of diag(-1,1,-1)
            # = (0,0,1,0)), :332 rotations_local[flip] = flip_matrix (x) rotations_local[flip],
            # :360 xyzs_local[flip, flip_axis] *= -1
            fq = torch.tensor([0.0, 0.0, 1.0, 0.0], device=m.xyz.device).expand(n, 4)
            rot_local = torch.where(flip[:, None], quaternion_raw_multiply(fq, rot_local), rot_local)
            xyz_local = torch.where(flip[:, None] & (torch.arange(3, device=m.xyz.device) == 1)[None], -m.xyz, m.xyz)
        if p is None:
            xyzs.append(m.xyz)
            rots.append(rot_local)
            feats.append(torch.cat((m.features_dc, m.features_rest), dim=1))
        else:
            dev = m.xyz.device   # tensors keep their graph: autograd through the poses works (training)
            obj_rot = torch.as_tensor(p[0], dtype=torch.float32).to(dev).reshape(1, 4).expand(n, -1)
            obj_trans = torch.as_tensor(p[1], dtype=torch.float32).to(dev).reshape(1, 3).expand(n, -1)
            R = quaternion_to_matrix(obj_rot)
            xyzs.append(torch.einsum('bij, bj -> bi', R, xyz_local) + obj_trans)
            rots.append(torch.nn.functional.normalize(quaternion_raw_multiply(obj_rot, rot_local)))
            base = idft(p[2], m.features_dc.shape[1])[0].to(m.xyz.device)
            dc = torch.sum(m.features_dc * base[..., None], dim=1, keepdim=True)
            feats.append(torch.cat([dc, m.features_rest], dim=1))
    return (torch.cat(xyzs, 0), torch.cat(scales, 0), torch.cat(rots, 0), torch.cat(opas, 0),
            torch.cat(feats, 0))
