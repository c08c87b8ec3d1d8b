"""Pure-PyTorch CPU splat -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Second, independent restatement of the reference rasterizer (the first is gs_oracle.c):
vectorised preprocess, ``torch.sort`` on the 64-bit (tile | depth-bits) key, per-tile
front-to-back alpha blend with the reference's exact accept/reject rules.  Written from the
maths, not from the CUDA, so that agreement with gs_oracle.c is evidence
that both decode the reference's conventions correctly.  Uses:

* ``bench.py`` ``cpu_baseline`` leg -- BASELINE.json north_star asks for "a pure-PyTorch CPU
  splat baseline timed on the host cores";
* gradient reference: run in float64 with autograd (all accept/reject decisions are piecewise
  constant, so autograd through the taken branches is the exact gradient);
* the PSNR-parity toy fit of config 5.

Reference lines restated: cuda_rasterizer/forward.cu:20-71 (SH), :74-113 (EWA), :118-152 (Σ),
:155-256 (preprocess), :340-467 (blend); rasterizer_impl.cu:70-138 (keys, ranges);
auxiliary.h:41-56,139-164.  Never imported by the product path.
"""
import math

import torch

BLOCK = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
      0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, shs, dirs):
    """forward.cu:20-71 without the +0.5/clamp.  shs [P,M,3], dirs [P,3] unit."""
    res = C0 * shs[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5]
                   + C2[2] * (2.0 * zz - xx - yy) * shs[:, 6] + C2[3] * xz * shs[:, 7]
                   + C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + C3[0] * y * (3.0 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10]
                       + C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11]
                       + C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
                       + C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13]
                       + C3[5] * z * (xx - yy) * shs[:, 14] + C3[6] * x * (xx - 3.0 * yy) * shs[:, 15])
    return res


def quat_to_rotmat(q):
    """Standard rotation matrix of quaternion (r,x,y,z), NOT normalised (forward.cu:127)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def preprocess(means3D, opacities, *, image_height, image_width, tanfovx, tanfovy, viewmatrix,
               projmatrix, campos, sh_degree, scale_modifier=1.0, shs=None, colors_precomp=None,
               scales=None, rotations=None, cov3D_precomp=None):
    H, W = int(image_height), int(image_width)
    dt = means3D.dtype
    P = means3D.shape[0]
    view = viewmatrix.to(dt)
    proj = projmatrix.to(dt)
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], dim=1)
    p_hom = hom @ proj
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]
    p_view = (hom @ view)[:, :3]
    visible = p_view[:, 2] > 0.2

    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4],
                             c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    else:
        Rm = quat_to_rotmat(rotations)
        L = Rm * (scale_modifier * scales)[:, None, :]
        Sigma = L @ L.transpose(1, 2)

    focal_y = H / (2.0 * tanfovy)
    focal_x = W / (2.0 * tanfovx)
    tz = p_view[:, 2]
    tz_safe = torch.where(visible, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz_safe, -limx, limx) * tz_safe
    ty = torch.clamp(p_view[:, 1] / tz_safe, -limy, limy) * tz_safe
    zeros = torch.zeros_like(tz)
    J = torch.stack([focal_x / tz_safe, zeros, -(focal_x * tx) / (tz_safe * tz_safe),
                     zeros, focal_y / tz_safe, -(focal_y * ty) / (tz_safe * tz_safe)], 1).reshape(-1, 2, 3)
    Wm = view[:3, :3].t()          # math rotation: t = Wm @ p + trans
    JW = J @ Wm
    cov = JW @ Sigma @ JW.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c_ = cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    ok = visible & (det != 0)
    det_safe = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c_ / det_safe, -b / det_safe, a / det_safe], 1)
    mid = 0.5 * (a + c_)
    disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    lam = torch.maximum(mid + disc, mid - disc)
    radius = torch.ceil(3.0 * torch.sqrt(torch.clamp(lam, min=0.0))).detach()
    # ndc2Pix is evaluated in fp64 in the reference (auxiliary.h:41-44)
    px = (((p_proj[:, 0].double() + 1.0) * W - 1.0) * 0.5).to(dt)
    py = (((p_proj[:, 1].double() + 1.0) * H - 1.0) * 0.5).to(dt)
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    with torch.no_grad():
        r = torch.where(ok, radius, torch.zeros_like(radius))
        minx = torch.clamp(torch.trunc((px - r) / BLOCK), 0, gx).long()
        miny = torch.clamp(torch.trunc((py - r) / BLOCK), 0, gy).long()
        maxx = torch.clamp(torch.trunc((px + r + (BLOCK - 1)) / BLOCK), 0, gx).long()
        maxy = torch.clamp(torch.trunc((py + r + (BLOCK - 1)) / BLOCK), 0, gy).long()
        tiles = (maxx - minx) * (maxy - miny)
        ok = ok & (tiles > 0)
        tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
        radii = torch.where(ok, r, torch.zeros_like(r)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos.to(dt)[None, :]
        d = d / torch.sqrt((d * d).sum(1, keepdim=True))
        raw = sh_to_rgb(sh_degree, shs, d) + 0.5
        clamped = (raw < 0).detach()
        rgb = torch.clamp(raw, min=0.0)
    else:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    return dict(radii=radii, means2D=torch.stack([px, py], 1), depths=p_view[:, 2], conic=conic,
                opacity=opacities.reshape(-1), rgb=rgb, clamped=clamped, tiles_touched=tiles,
                rect=(minx, miny, maxx, maxy), grid=(gx, gy), visible=ok)


def bin_tiles(pre):
    """duplicateWithKeys + stable sort + identifyTileRanges (rasterizer_impl.cu:70-138)."""
    minx, miny, maxx, maxy = pre["rect"]
    gx, gy = pre["grid"]
    tiles = pre["tiles_touched"]
    vis = torch.nonzero(tiles > 0).squeeze(1)
    counts = tiles[vis]
    R = int(counts.sum())
    gid = torch.repeat_interleave(vis, counts)
    starts = torch.cumsum(counts, 0) - counts
    local = torch.arange(R) - torch.repeat_interleave(starts, counts)
    w = (maxx - minx)[gid]
    ty = miny[gid] + local // torch.clamp(w, min=1)
    tx = minx[gid] + local % torch.clamp(w, min=1)
    tile = ty * gx + tx
    dbits = pre["depths"].detach().to(torch.float32).view(torch.int32).long()[gid]
    keys = (tile << 32) | dbits
    keys_sorted, order = torch.sort(keys, stable=True)
    point_list = gid[order]
    tile_sorted = keys_sorted >> 32
    T = gx * gy
    cnt = torch.bincount(tile_sorted, minlength=T)
    end = torch.cumsum(cnt, 0)
    start = end - cnt
    ranges = torch.stack([torch.where(cnt > 0, start, torch.zeros_like(start)),
                          torch.where(cnt > 0, end, torch.zeros_like(end))], 1)
    return dict(num_rendered=R, keys_sorted=keys_sorted, point_list=point_list, ranges=ranges)


def render_tiles(pre, binning, *, image_height, image_width, bg, semantics=None,
                 tile_subset=None):
    """Per-tile blend (forward.cu:340-467), differentiable w.r.t. everything in ``pre``."""
    H, W = int(image_height), int(image_width)
    gx, gy = pre["grid"]
    dt = pre["means2D"].dtype
    S = 0 if semantics is None else semantics.shape[1]
    color = torch.zeros(3, H, W, dtype=dt)
    depth = torch.zeros(1, H, W, dtype=dt)
    alpha_out = torch.zeros(1, H, W, dtype=dt)
    sem = torch.zeros(S, H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    bg = bg.to(dt)
    color = color + bg[:, None, None]
    ranges = binning["ranges"]
    plist = binning["point_list"]
    ly, lx = torch.meshgrid(torch.arange(BLOCK), torch.arange(BLOCK), indexing="ij")
    tiles = range(gx * gy) if tile_subset is None else tile_subset
    out_c, out_d, out_a, out_s = [], [], [], []
    for t in tiles:
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        ty, tx = divmod(t, gx)
        y0, x0 = ty * BLOCK, tx * BLOCK
        hh, ww = min(BLOCK, H - y0), min(BLOCK, W - x0)
        if r1 <= r0:
            continue
        ids = plist[r0:r1]
        pxs = (x0 + lx[:hh, :ww]).reshape(-1).to(dt)
        pys = (y0 + ly[:hh, :ww]).reshape(-1).to(dt)
        xy = pre["means2D"][ids]
        con = pre["conic"][ids]
        op = pre["opacity"][ids]
        dx = xy[:, 0:1] - pxs[None, :]
        dy = xy[:, 1:2] - pys[None, :]
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        alpha = torch.clamp(op[:, None] * torch.exp(power), max=0.99)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        a = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tinc = torch.cumprod(1 - a, dim=0)
        alive = Tinc >= 0.0001
        contrib = valid & alive
        Texc = torch.cat([torch.ones(1, a.shape[1], dtype=dt), Tinc[:-1]], 0)
        wgt = torch.where(contrib, a * Texc, torch.zeros_like(a))
        T_final = torch.prod(torch.where(contrib, 1 - a, torch.ones_like(a)), dim=0)
        C = wgt.t() @ pre["rgb"][ids]                      # [pix,3]
        D = wgt.t() @ pre["depths"][ids]
        A = wgt.sum(0)
        idx = torch.arange(1, ids.shape[0] + 1)[:, None]
        nc = torch.where(contrib, idx, torch.zeros_like(idx)).max(0).values
        col = (C + T_final[:, None] * bg[None, :]).t().reshape(3, hh, ww)
        color = _paste(color, col, y0, x0)
        depth = _paste(depth, D.reshape(1, hh, ww), y0, x0)
        alpha_out = _paste(alpha_out, A.reshape(1, hh, ww), y0, x0)
        if S:
            sem = _paste(sem, (wgt.t() @ semantics[ids]).t().reshape(S, hh, ww), y0, x0)
        n_contrib[y0:y0 + hh, x0:x0 + ww] = nc.reshape(hh, ww).to(torch.int32)
    return dict(color=color, depth=depth, alpha=alpha_out, semantic=sem, n_contrib=n_contrib)


def _paste(img, patch, y0, x0):
    """Functional (autograd-safe) paste of a [C,h,w] patch."""
    _, h, w = patch.shape
    pad = torch.nn.functional.pad(patch, (x0, img.shape[2] - x0 - w, y0, img.shape[1] - y0 - h))
    mask = torch.zeros(1, img.shape[1], img.shape[2], dtype=torch.bool)
    mask[:, y0:y0 + h, x0:x0 + w] = True
    return torch.where(mask, pad, img)


def rasterize(means3D, opacities, *, image_height, image_width, tanfovx, tanfovy, bg,
              scale_modifier, viewmatrix, projmatrix, sh_degree, campos, shs=None,
              colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
              semantics=None, tile_subset=None, **_unused):
    """End-to-end CPU splat with the op's semantics.  Returns dict incl. the 5 op outputs."""
    pre = preprocess(means3D, opacities, image_height=image_height, image_width=image_width,
                     tanfovx=tanfovx, tanfovy=tanfovy, viewmatrix=viewmatrix,
                     projmatrix=projmatrix, campos=campos, sh_degree=sh_degree,
                     scale_modifier=scale_modifier, shs=shs, colors_precomp=colors_precomp,
                     scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    binning = bin_tiles(pre)
    out = render_tiles(pre, binning, image_height=image_height, image_width=image_width, bg=bg,
                       semantics=semantics, tile_subset=tile_subset)
    out.update(radii=pre["radii"], num_rendered=binning["num_rendered"], pre=pre, binning=binning)
    return out


def psnr(a, b):
    mse = torch.mean((a - b) ** 2)
    return float(10.0 * math.log10(1.0 / max(float(mse), 1e-20)))
