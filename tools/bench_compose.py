"""Per-frame cost of the scene-graph composition in front of the op:
(a) the reference's way -- PyTorch activations / rigid transforms / IDFT / cat, then the classic op;
(b) gaussianrpg_amd.composed.ComposedRasterizer (fused into preprocess).
Scene: scene-002-like, 1.9 M background Gaussians + 10 actors of 10 k.
Prints one JSON line (synchronize-bracketed medians over the 200-pose drive)."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from gaussianrpg_amd.composed import ActorPose, ComposedRasterizer, ModelParams, idft_weights
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(2)
NB, NA, PA, F = 1_900_000, 10, 10_000, 5
sc = hz.street_scene(NB, seed=2)
models = [ModelParams(sc.means3D, torch.log(sc.scales), sc.rotations * 1.7,
                      torch.log(sc.opacity.clamp(1e-4, 1 - 1e-4) / (1 - sc.opacity.clamp(1e-4, 1 - 1e-4))),
                      sc.shs[:, :1].contiguous(), sc.shs[:, 1:].contiguous())]
for k in range(NA):
    models.append(ModelParams((torch.rand(PA, 3, generator=g) - 0.5) * torch.tensor([4.5, 1.6, 2.0]),
                              math.log(0.05) + 0.5 * torch.randn(PA, 3, generator=g),
                              torch.randn(PA, 4, generator=g), 1.0 + 2.0 * torch.randn(PA, 1, generator=g),
                              0.5 * torch.randn(PA, F, 3, generator=g), 0.15 * torch.randn(PA, 3, 3, generator=g)))
models = [ModelParams(*(t.to(dev) for t in m[:6])) for m in models]


def poses_at(f):
    out = [None]
    for k in range(NA):
        a = 0.05 * k + 0.002 * f
        out.append(ActorPose([math.cos(a), 0.0, math.sin(a), 0.0], [-12.0 + 2.5 * k, 0.8, 10.0 + 8.0 * k + 0.5 * f],
                             0.1 + 0.004 * f))
    return out


def qmul(a, b):
    aw, ax, ay, az = torch.unbind(a, -1); bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw*bw-ax*bx-ay*by-az*bz, aw*bx+ax*bw+ay*bz-az*by, aw*by-ax*bz+ay*bw+az*bx,
                        aw*bz+ax*by-ay*bx+az*bw), -1)


def qmat(r):
    q = r / torch.sqrt((r * r).sum(1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y), 2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x),
                        2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)], 1).reshape(-1, 3, 3)


def torch_compose(poses):
    """the reference's getters (street_gaussian_model.py:296-453) on the device"""
    xyz, sca, rot, opa, fea = [models[0].xyz], [torch.exp(models[0].scaling)], \
        [torch.nn.functional.normalize(models[0].rotation)], [torch.sigmoid(models[0].opacity)], \
        [torch.cat((models[0].features_dc, models[0].features_rest), 1)]
    loc_x = torch.cat([m.xyz for m in models[1:]]); loc_r = torch.cat([torch.nn.functional.normalize(m.rotation) for m in models[1:]])
    orot = torch.cat([torch.tensor(p.obj_rot, device=dev).expand(PA, -1) for p in poses[1:]])
    otr = torch.cat([torch.tensor(p.obj_trans, device=dev).expand(PA, -1) for p in poses[1:]])
    xyz.append(torch.einsum('bij,bj->bi', qmat(orot), loc_x) + otr)
    rot.append(torch.nn.functional.normalize(qmul(orot, loc_r)))
    for m, p in zip(models[1:], poses[1:]):
        sca.append(torch.exp(m.scaling)); opa.append(torch.sigmoid(m.opacity))
        base = torch.tensor(idft_weights(p.fourier_time, F), device=dev)
        fea.append(torch.cat([torch.sum(m.features_dc * base[..., None], 1, keepdim=True), m.features_rest], 1))
    return torch.cat(xyz), torch.cat(sca), torch.cat(rot), torch.cat(opa), torch.cat(fea)


cams = [hz.trajectory_camera(f, device=dev) for f in range(60)]
rss = [GaussianRasterizationSettings(**hz.settings_kwargs(c, 1)) for c in cams]


def run_ref(f):
    with torch.no_grad():
        x, s, r, o, sh = torch_compose(poses_at(f))
        return GaussianRasterizer(rss[f])(means3D=x, means2D=None, opacities=o, shs=sh, scales=s, rotations=r)[0]


def run_fused(f):
    return ComposedRasterizer(rss[f])(models, poses_at(f))[0]


a = hz.time_frames(run_ref, 60)
b = hz.time_frames(run_fused, 60)

# training step (forward + backward to the RAW parameters): PyTorch composition with autograd + the
# classic op, against the fused op's own backward (grpg_backward_composed)
train_models = [ModelParams(*(t.clone().requires_grad_(True) for t in m[:6])) for m in models]


def train_ref(f):
    global models
    keep, models = models, train_models
    try:
        x, s, r, o, sh = torch_compose(poses_at(f))
    finally:
        models = keep
    m2d = torch.zeros(x.shape[0], 3, device=dev, requires_grad=True)
    c, _, d, al, _ = GaussianRasterizer(rss[f])(means3D=x, means2D=m2d, opacities=o, shs=sh, scales=s, rotations=r)
    (c.mean() + 0.1 * d.mean() + al.mean()).backward()


def train_fused(f):
    P = sum(m.xyz.shape[0] for m in train_models)
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    c, _, d, al = ComposedRasterizer(rss[f])(train_models, poses_at(f), means2D=m2d)
    (c.mean() + 0.1 * d.mean() + al.mean()).backward()


def zero_grads():
    for m in train_models:
        for t in m[:6]:
            t.grad = None


def time_train(fn, n=24):
    ts = []
    for k in range(n):
        zero_grads()
        torch.cuda.synchronize(); t0 = time.time()
        fn(k)
        torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    ts = sorted(ts[2:])
    return {"iterations": len(ts), "median_ms": ts[len(ts) // 2], "p95_ms": ts[int(0.95 * len(ts))]}


ta = time_train(train_ref)
tb = time_train(train_fused)


# render_all on the scene graph (street_gaussian_renderer.py:13-40): the reference composes and renders three
# times per frame -- all models, the background alone, the objects alone (set_visibility + parse_camera + the
# getters + the op, each time) -- against ONE composed layered forward (grpg_forward_composed_layers)
def torch_compose_subset(poses, keep_bg, keep_obj):
    global models
    full, allposes = models, poses
    try:
        if keep_bg and not keep_obj:      # background alone: no actor arithmetic at all
            m = full[0]
            return (m.xyz, torch.exp(m.scaling), torch.nn.functional.normalize(m.rotation), torch.sigmoid(m.opacity),
                    torch.cat((m.features_dc, m.features_rest), 1))
        x, s, r, o, sh = torch_compose(allposes)
        if keep_bg:
            return x, s, r, o, sh
        nb = full[0].xyz.shape[0]         # objects alone: the actors' slices of the composition
        return x[nb:], s[nb:], r[nb:], o[nb:], sh[nb:]
    finally:
        models = full


white = torch.ones(3, device=dev)
rss_white = [GaussianRasterizationSettings(**hz.settings_kwargs(c, 1, bg=white)) for c in cams]


def render_all_ref(f):
    with torch.no_grad():
        outs = []
        for (kb, ko, rs) in ((True, True, rss[f]), (True, False, rss_white[f]), (False, True, rss_white[f])):
            x, s, r, o, sh = torch_compose_subset(poses_at(f), kb, ko)
            outs.append(GaussianRasterizer(rs)(means3D=x, means2D=None, opacities=o, shs=sh, scales=s, rotations=r)[0])
        return outs


def render_all_fused(f):
    return ComposedRasterizer(rss[f]).forward_layers(models, poses_at(f))["color"]


ra = hz.time_frames(render_all_ref, 40)
rb = hz.time_frames(render_all_fused, 40)
print(json.dumps({"what": "scene-graph composition + forward op, 1.9 M background + 10 actors x 10 k, 1920x1280",
                  "torch_composition_then_op_ms": a, "fused_composed_op_ms": b,
                  "render_all": {"what": "the evaluation path's three renders per frame (all / background / objects), "
                                         "each with its own composition, vs one composed layered forward",
                                 "torch_composition_and_three_ops_ms": ra, "fused_composed_layers_ms": rb},
                  "train_step": {"what": "forward + backward down to the raw parameters (loss = mean colour + "
                                         "0.1 mean depth + mean alpha), synchronize-bracketed wall time",
                                 "torch_composition_autograd_plus_op_ms": ta, "fused_composed_op_ms": tb}}))
