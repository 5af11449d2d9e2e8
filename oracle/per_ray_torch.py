"""Pure-PyTorch CPU restatement of the 3DGUT path with autograd: "oracle flavour (b)" of SURVEY §7.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  "parity unpinned": the
reference ships no fixtures for this path; this file is an independent second restatement used to
cross-check the C oracle (gut_oracle.c) — forward values and, through autograd in float64, the
hand-derived backward of the reference (models/gaussianParticles.cuh:480-738).

Two entry points:
  * render_tiled(...)   — composites every tile against the tile's own depth-sorted list
                          (semantics identical to K6; lists come from the structural pipeline);
  * render_per_ray(...) — brute force, no tiles: every pixel against all projected Gaussians sorted
                          by camera-space z (BASELINE.json configs[0]; the timed CPU baseline).
Both are differentiable w.r.t. positions / rotation / scale / density / SH features and replicate
the reference's backward conventions:
  - alpha clamp min(0.99, .) is straight-through (gaussianParticles.cuh:611-629 are not masked);
  - SH view direction carries no gradient (gaussianParticles.slang:307-318 `no_diff`);
  - the hit-distance output uses a detached transmittance (quirk 1 of SURVEY §8a); its gradient
    w.r.t. the canonical ray is only diagonal in the reference, so gradient cross-checks are made
    with dist_grad == 0.
"""
import math

import torch

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
MIN_RESPONSE = 0.0113
T_MIN = 1e-4

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def quat_rows(q):
    """rows r0,r1,r2 of rotationT from wxyz quaternions [N,4] -> [N,3,3] (transforms.slang:22-39)."""
    w, x, y, z = q.unbind(-1)
    xx, yy, zz, xy, xz, yz, rx, ry, rz = x * x, y * y, z * z, x * y, x * z, y * z, w * x, w * y, w * z
    r0 = torch.stack([1 - 2 * (yy + zz), 2 * (xy + rz), 2 * (xz - ry)], -1)
    r1 = torch.stack([2 * (xy - rz), 1 - 2 * (xx + zz), 2 * (yz + rx)], -1)
    r2 = torch.stack([2 * (xz + ry), 2 * (yz - rx), 1 - 2 * (xx + yy)], -1)
    return torch.stack([r0, r1, r2], -2)


def sh_basis(deg, d):
    x, y, z = d.unbind(-1)
    Y = [torch.full_like(x, _C0)]
    if deg > 0:
        Y += [-_C1 * y, _C1 * z, -_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        Y += [_C2[0] * xy, _C2[1] * yz, _C2[2] * (2 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy)]
    if deg > 2:
        Y += [_C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy),
              _C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy),
              _C3[6] * x * (xx - 3 * yy)]
    return torch.stack(Y, -1)  # [N, (deg+1)^2]


def precompute_features(pos, sph48, cam_pos, deg):
    """per-Gaussian view-dependent colour, unclamped, +0.5 (gutProjector.cuh:304-310)."""
    d = (pos - cam_pos).detach()
    d = d / d.norm(dim=-1, keepdim=True)
    Y = sh_basis(deg, d)
    nc = (deg + 1) ** 2
    return (Y.unsqueeze(-1) * sph48.view(-1, 16, 3)[:, :nc]).sum(1) + 0.5


def pose_matrices(tq, dtype):
    """world->sensor (R,t) and sensor->world (R^T, cam_pos) from [t, q_xyzw]."""
    t = torch.as_tensor(tq[:3], dtype=dtype)
    x, y, z, w = [float(v) for v in tq[3:7]]
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=dtype)
    return R, t, R.T, -(R.T @ t)


def world_rays(ray_ori, ray_dir, Rinv, cam_pos):
    o = ray_ori.reshape(-1, 3) @ Rinv.T + cam_pos
    d = ray_dir.reshape(-1, 3) @ Rinv.T
    return o, d


# generalised Gaussian kernels (render.particle_kernel_degree): resp = exp(-S_n d2^(n/2)), S_n = 4.5 / 3^n as the reference writes
# them (kernels/cuda/models/gaussianParticles.cuh:256-306, slang/models/gaussianParticles.slang:119-164); n = 0 is the linear hat
_KERNEL_S = {8: 0.000685871056241, 5: 0.0185185185185, 4: 0.0555555555556, 3: 0.166666666667, 2: 0.5, 1: 1.5, 0: 0.329630334487}


def kernel_response(d2, degree=2):
    """The kernel's response as a plain differentiable expression (true derivative under autograd: what the reference's slang
    autodiff gives the sorted variant)."""
    if degree not in _KERNEL_S:
        degree = 2   # particleResponse<>'s default case
    sn = _KERNEL_S[degree]
    if degree == 0:
        return (1.0 - sn * torch.sqrt(d2)).clamp(min=0.0)
    if degree == 2:
        return torch.exp(-0.5 * d2)
    return torch.exp(-sn * d2 ** (0.5 * degree))


class _ReferenceResponse(torch.autograd.Function):
    """Same forward; the backward is particleResponseGrd<n> AS THE REFERENCE WRITES IT for its unsorted backward
    (gaussianParticles.cuh:211-254).  For every degree but 1 that is the true derivative; for degree 1 it multiplies by sqrt(d2)
    where the derivative of exp(s sqrt(d2)) divides by it (:248-252)."""

    @staticmethod
    def forward(ctx, d2, degree):
        resp = kernel_response(d2, degree)
        ctx.save_for_backward(d2, resp)
        ctx.degree = degree if degree in _KERNEL_S else 2
        return resp

    @staticmethod
    def backward(ctx, g):
        d2, resp = ctx.saved_tensors
        n = ctx.degree
        sn = _KERNEL_S[n]
        if n == 8:
            out = -sn * 4.0 * d2 * d2 * d2 * resp * g
        elif n == 5:
            out = -sn * 2.5 * d2 * torch.sqrt(d2) * resp * g
        elif n == 4:
            out = -sn * 2.0 * d2 * resp * g
        elif n == 3:
            out = -sn * 1.5 * torch.sqrt(d2) * resp * g
        elif n == 1:
            out = -sn * 0.5 * torch.sqrt(d2) * resp * g
        elif n == 0:
            out = torch.where(resp > 0, 0.5 * -sn / torch.sqrt(d2) * g, torch.zeros_like(g))
        else:
            out = -0.5 * resp * g
        return out, None


def _composite_block(o, d, tmin, tmax, mu, rows, s, sigma, feat, state, kernel_degree=2, reference_response_grad=False):
    """One block of P rays against L depth-ordered Gaussians; state = (T, alive, rgb, dist, hits)."""
    T_c, alive_c, rgb, dist, hits = state
    gposc = o[:, None, :] - mu[None]
    gposcr = torch.einsum("lij,plj->pli", rows, gposc)
    gro = gposcr / s
    grdu = torch.einsum("lij,pj->pli", rows, d) / s
    grd = grdu / grdu.norm(dim=-1, keepdim=True)
    c = torch.cross(grd, gro, dim=-1)
    d2 = (c * c).sum(-1)
    resp = _ReferenceResponse.apply(d2, kernel_degree) if reference_response_grad else kernel_response(d2, kernel_degree)
    a = resp * sigma[None, :]
    alpha = a + (a.clamp(max=ALPHA_MAX) - a).detach()
    p = -(grd * gro).sum(-1)
    hit_t = (s * grd * p[..., None]).norm(dim=-1)
    accept = (resp > MIN_RESPONSE) & (alpha > ALPHA_MIN) & (hit_t > tmin[:, None]) & (hit_t < tmax[:, None])
    a_eff = torch.where(accept, alpha, torch.zeros_like(alpha))
    one_m = 1 - a_eff
    cp = torch.cumprod(one_m, dim=1)
    T_before = T_c[:, None] * torch.cat([torch.ones_like(cp[:, :1]), cp[:, :-1]], dim=1)
    proc = accept & alive_c[:, None] & (T_before.detach() >= T_MIN)
    w = torch.where(proc, a_eff * T_before, torch.zeros_like(a_eff))
    pos_w = w > 0
    rgb = rgb + (torch.where(pos_w, w, torch.zeros_like(w))[..., None] * feat.clamp(min=0)[None]).sum(1)
    w_det = torch.where(proc, a_eff * T_before.detach(), torch.zeros_like(a_eff))
    dist = dist + (w_det * hit_t).sum(1)
    hits = hits + pos_w.sum(1).to(hits.dtype)
    T_new = T_c * torch.where(proc, one_m, torch.ones_like(one_m)).prod(dim=1)
    alive_new = alive_c & (T_new.detach() >= T_MIN)
    return T_new, alive_new, rgb, dist, hits


def _ray_limits(o, d):
    """slab test against the +-1e6 box, utils/bounding_box.h:88-134 (only the alive test matters)."""
    big = 1e6
    t0 = (-big - o) / d
    t1 = (big - o) / d
    tmin = torch.minimum(t0, t1).max(dim=-1).values.clamp(min=0)
    tmax = torch.maximum(t0, t1).min(dim=-1).values
    return tmin, tmax


def render_tiled(params, tq, W, H, ray_ori, ray_dir, tile_ranges, sorted_ids, sh_degree=3, dtype=torch.float64,
                 chunk=128, kernel_degree=2, reference_response_grad=False):
    """params: dict of tensors positions[N,3], rotation[N,4], scale[N,3], density[N,1], features[N,48].
    Returns rgba [H,W,4], dist [H,W], hits [H,W] (torch, differentiable)."""
    R, t, Rinv, cam_pos = pose_matrices(tq, dtype)
    pos, rot, scl, dns, sph = (params[k].to(dtype) for k in ("positions", "rotation", "scale", "density", "features"))
    feat_all = precompute_features(pos, sph, cam_pos, sh_degree)
    rows_all = quat_rows(rot)
    o_all, d_all = world_rays(torch.as_tensor(ray_ori, dtype=dtype), torch.as_tensor(ray_dir, dtype=dtype), Rinv, cam_pos)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rgba = torch.zeros(H * W, 4, dtype=dtype)
    dist_o = torch.full((H * W,), 1e6, dtype=dtype)
    hits_o = torch.zeros(H * W, dtype=dtype)
    ids_t = torch.as_tensor(sorted_ids.astype("int64"))
    for tile in range(gx * gy):
        ty, tx = divmod(tile, gx)
        ys = torch.arange(ty * 16, min(ty * 16 + 16, H))
        xs = torch.arange(tx * 16, min(tx * 16 + 16, W))
        pix = (ys[:, None] * W + xs[None, :]).reshape(-1)
        o, d = o_all[pix], d_all[pix]
        tmin, tmax = _ray_limits(o, d)
        P = pix.numel()
        state = (torch.ones(P, dtype=dtype), tmax > tmin, torch.zeros(P, 3, dtype=dtype), torch.zeros(P, dtype=dtype),
                 torch.zeros(P, dtype=dtype))
        valid = state[1].clone()
        b, e = int(tile_ranges[tile][0]), int(tile_ranges[tile][1])
        for c0 in range(b, e, chunk):
            ids = ids_t[c0:min(c0 + chunk, e)]
            ids = ids[ids != 0xFFFFFFFF]
            if ids.numel() == 0 or not bool(state[1].any()):
                break
            state = _composite_block(o, d, tmin, tmax, pos[ids], rows_all[ids], scl[ids], dns[ids, 0], feat_all[ids], state,
                                     kernel_degree, reference_response_grad)
        T, _, rgb, dist, hits = state
        out = torch.cat([rgb, (1 - T)[:, None]], dim=1)
        rgba = rgba.index_put((pix[valid],), out[valid])
        dist_o = dist_o.index_put((pix[valid],), dist[valid])
        hits_o = hits_o.index_put((pix[valid],), hits[valid])
    return rgba.view(H, W, 4), dist_o.view(H, W), hits_o.view(H, W)


# ---------------------------------------------------------------------------------------------
# K1 in torch (vectorised): UT projection, conic/extent, cull flags.  Tolerance-level cross-check of
# the C oracle's projection and the culling rule of the brute-force baseline.
# ---------------------------------------------------------------------------------------------
def _project_points(cam, R, t, W, H, pts, margin=0.1):
    p = pts @ R.T + t
    dtype = p.dtype
    pp = torch.tensor(cam["principal_point"], dtype=dtype)
    fl = torch.tensor(cam["focal_length"], dtype=dtype)
    if cam["model"] == "pinhole":
        rad = list(cam.get("radial", [])) + [0.0] * 6
        tan = list(cam.get("tangential", [0.0, 0.0]))
        thp = list(cam.get("thin_prism", [0.0] * 4))
        z = p[..., 2]
        front = z > 0
        zs = torch.where(front, z, torch.ones_like(z))
        u, v = p[..., 0] / zs, p[..., 1] / zs
        r2 = u * u + v * v
        a1, a2, a3 = 2 * u * v, r2 + 2 * u * u, r2 + 2 * v * v
        icd = (1 + r2 * (rad[0] + r2 * (rad[1] + r2 * rad[2]))) / (1 + r2 * (rad[3] + r2 * (rad[4] + r2 * rad[5])))
        dx = tan[0] * a1 + tan[1] * a2 + r2 * (thp[0] + r2 * thp[1])
        dy = tan[0] * a3 + tan[1] * a1 + r2 * (thp[2] + r2 * thp[3])
        ok_rad = (icd > 0.8) & (icd < 1.2)
        x_ok = (icd * u + dx) * fl[0] + pp[0]
        y_ok = (icd * v + dy) * fl[1] + pp[1]
        k = math.hypot(W, H) / torch.sqrt(r2.clamp(min=1e-30))
        x = torch.where(ok_rad, x_ok, k * u + pp[0])
        y = torch.where(ok_rad, y_ok, k * v + pp[1])
        x = torch.where(front, x, torch.zeros_like(x))
        y = torch.where(front, y, torch.zeros_like(y))
        ok = front & ok_rad
    else:
        rad = list(cam.get("radial", [])) + [0.0] * 4
        rho = p[..., :2].norm(dim=-1).clamp(min=1.1920929e-07)
        th_full = torch.atan2(rho, p[..., 2])
        th = th_full.clamp(max=cam["max_angle"])
        t2 = th * th
        poly = ((rad[3] * t2 + rad[2]) * t2 + rad[1]) * t2 + rad[0]
        delta = th * (poly * t2 + 1) / rho
        x = fl[0] * p[..., 0] * delta + pp[0]
        y = fl[1] * p[..., 1] * delta + pp[1]
        ok = th < cam["max_angle"]
    inside = (x > -W * margin) & (y > -H * margin) & (x < W + W * margin) & (y < H + H * margin)
    return torch.stack([x, y], -1), ok & inside


def project(cam, tq, W, H, params, dtype=torch.float64):
    """Returns dict(valid[N] bool, center[N,2], conic_opacity[N,4], extent[N,2], depth[N])."""
    R, t, Rinv, cam_pos = pose_matrices(tq, dtype)
    pos, rot, scl, dns = (params[k].detach().to(dtype) for k in ("positions", "rotation", "scale", "density"))
    rows = quat_rows(rot)
    z = pos @ R[2] + t[2]
    delta = math.sqrt(3.0) * scl[:, :, None] * rows  # [N,3(axis),3]
    pts = torch.cat([pos[:, None], pos[:, None] + delta, pos[:, None] - delta], dim=1)  # [N,7,3]
    xy, ok = _project_points(cam, R, t, W, H, pts)
    center = xy[:, 1:].sum(1) / 6.0
    e0 = xy[:, 0] - center
    ei = xy[:, 1:] - center[:, None]
    cov = torch.stack([2 * e0[:, 0] ** 2 + (ei[..., 0] ** 2).sum(1) / 6, 2 * e0[:, 0] * e0[:, 1] + (ei[..., 0] * ei[..., 1]).sum(1) / 6,
                       2 * e0[:, 1] ** 2 + (ei[..., 1] ** 2).sum(1) / 6], -1)
    dx, dy, dz = cov[:, 0] + 0.3, cov[:, 1], cov[:, 2] + 0.3
    ddet = dx * dz - dy * dy
    conic = torch.stack([dz / ddet, -dy / ddet, dx / ddet], -1)
    op = dns[:, 0] * torch.sqrt(((cov[:, 0] * cov[:, 2] - cov[:, 1] ** 2) / ddet).clamp(min=0.000025))
    valid = (dns[:, 0] >= ALPHA_MIN) & (z >= 0.2) & ok.any(1) & (ddet != 0) & (op >= ALPHA_MIN)
    power = torch.log((op / ALPHA_MIN).clamp(min=1.0))
    ef = torch.sqrt(2 * power).clamp(max=3.33)
    mid = 0.5 * (dx + dz)
    radius = ef * torch.sqrt(mid + torch.sqrt((mid * mid - ddet).clamp(min=0.01)))
    extent = torch.minimum(ef[:, None] * torch.sqrt(torch.stack([dx, dz], -1)), radius[:, None])
    return dict(valid=valid, center=center, conic_opacity=torch.cat([conic, op[:, None]], -1), extent=extent, depth=z)


def tile_footprints(pr, W, H, tile=16):
    """Second, vectorised float64 restatement of K1's tile rules on top of project()'s output (independent code path from
    gut_oracle.c): the tile bounding box of gutProjector.cuh:32-43 (`pos - 0.5 +- extent`, floor / ceil, clamped to the grid) and
    the per-tile culling of :49-78 (power of the Gaussian at the rectangle point the reference picks, kept iff below
    ln(255 * opacity) = ln(opacity / min_alpha)), evaluated densely for every (Gaussian, tile) pair.
    Returns dict(in_box [N,T] bool, power [N,T], threshold [N], kept [N,T] bool, exact_min_power [N,T]) with T = tiles of the grid,
    row-major.  exact_min_power is the TRUE minimum of the same quadratic form over the tile rectangle (a convex function over a
    rectangle: zero if the mean is inside, else the smallest of the four clamped 1-D edge minima) — the reference evaluates the
    form at a feasible point of the rectangle, so power >= exact_min_power: a kept tile provably intersects the footprint."""
    c, e, co = pr["center"], pr["extent"], pr["conic_opacity"]
    dt = c.dtype
    gx, gy = (W + tile - 1) // tile, (H + tile - 1) // tile
    x0 = torch.floor((c[:, 0] - 0.5 - e[:, 0]) / tile).clamp(min=0, max=gx)
    y0 = torch.floor((c[:, 1] - 0.5 - e[:, 1]) / tile).clamp(min=0, max=gy)
    x1 = torch.ceil((c[:, 0] - 0.5 + e[:, 0]) / tile).clamp(min=0, max=gx)
    y1 = torch.ceil((c[:, 1] - 0.5 + e[:, 1]) / tile).clamp(min=0, max=gy)
    tx = torch.arange(gx, dtype=dt).repeat(gy)[None]               # [1,T] tile column
    ty = torch.arange(gy, dtype=dt).repeat_interleave(gx)[None]    # [1,T] tile row
    in_box = pr["valid"][:, None] & (tx >= x0[:, None]) & (tx < x1[:, None]) & (ty >= y0[:, None]) & (ty < y1[:, None])
    a, b, cc, op = (co[:, k][:, None] for k in range(4))
    mx, my = c[:, 0][:, None], c[:, 1][:, None]
    tminx, tminy = tile * tx, tile * ty
    tmaxx, tmaxy = tminx + tile, tminy + tile
    # --- the reference's point (gutProjector.cuh:57-74)
    offx, offy = tminx - mx, tminy - my
    lax, lay = (offx > 0).to(dt), (offy > 0).to(dt)
    nrx, nry = lax + (mx > tmaxx).to(dt), lay + (my > tmaxy).to(dt)
    px = torch.where(lax > 0, tminx, tmaxx)
    py = torch.where(lay > 0, tminy, tmaxy)
    dxx = torch.where(offx >= 0, torch.full_like(offx, float(tile)), torch.full_like(offx, -float(tile)))   # copysign(tile, offset)
    dyy = torch.where(offy >= 0, torch.full_like(offy, float(tile)), torch.full_like(offy, -float(tile)))
    fx_, fy_ = mx - px, my - py
    sx = nry * ((dxx * a * fx_ + dxx * b * fy_) / (tile * tile * a)).clamp(0, 1)
    sy = nrx * ((dyy * b * fx_ + dyy * cc * fy_) / (tile * tile * cc)).clamp(0, 1)
    qx, qy = mx - (px + sx * dxx), my - (py + sy * dyy)
    power = 0.5 * (a * qx * qx + cc * qy * qy) + b * qx * qy
    power = torch.where((nrx + nry) > 0, power, torch.zeros_like(power))
    thr = torch.log(op[:, 0] / ALPHA_MIN)
    kept = in_box & (power < thr[:, None])
    # --- exact minimum of q(d) = 0.5 (a dx^2 + c dy^2) + b dx dy over the rectangle, d = mean - point
    def q(dx_, dy_):
        return 0.5 * (a * dx_ * dx_ + cc * dy_ * dy_) + b * dx_ * dy_
    inside = (mx >= tminx) & (mx <= tmaxx) & (my >= tminy) & (my <= tmaxy)
    cands = []
    for xe in (tminx, tmaxx):        # vertical edges: x fixed, minimise over y in [tminy, tmaxy]: dq/ddy = c dy + b dx = 0
        dx_ = mx - xe
        ystar = torch.minimum(torch.maximum(my + b * dx_ / cc, tminy), tmaxy)
        cands.append(q(dx_, my - ystar))
    for ye in (tminy, tmaxy):        # horizontal edges
        dy_ = my - ye
        xstar = torch.minimum(torch.maximum(mx + b * dy_ / a, tminx), tmaxx)
        cands.append(q(mx - xstar, dy_))
    exact = torch.stack(cands, 0).min(0).values
    exact = torch.where(inside, torch.zeros_like(exact), exact)
    return dict(in_box=in_box, power=power, threshold=thr, kept=kept, exact_min_power=exact, grid=(gx, gy))


def render_per_ray(cam, tq, W, H, params, ray_ori, ray_dir, sh_degree=3, dtype=torch.float32, pix_chunk=4096,
                   gauss_chunk=512, pixel_subset=None):
    """Brute-force per-ray composite (no tiles): cull by the UT rule, sort by camera z, composite."""
    R, t, Rinv, cam_pos = pose_matrices(tq, dtype)
    pr = project(cam, tq, W, H, params, dtype)
    keep = torch.nonzero(pr["valid"]).squeeze(1)
    order = keep[torch.argsort(pr["depth"][keep], stable=True)]
    pos, rot, scl, dns, sph = (params[k].to(dtype) for k in ("positions", "rotation", "scale", "density", "features"))
    pos, rot, scl, dns, sph = pos[order], rot[order], scl[order], dns[order, 0], sph[order]
    feat = precompute_features(pos, sph, cam_pos, sh_degree)
    rows = quat_rows(rot)
    o_all, d_all = world_rays(torch.as_tensor(ray_ori, dtype=dtype), torch.as_tensor(ray_dir, dtype=dtype), Rinv, cam_pos)
    pix_all = torch.arange(H * W) if pixel_subset is None else torch.as_tensor(pixel_subset, dtype=torch.int64)
    outs = []
    for p0 in range(0, pix_all.numel(), pix_chunk):
        pix = pix_all[p0:p0 + pix_chunk]
        o, d = o_all[pix], d_all[pix]
        tmin, tmax = _ray_limits(o, d)
        P = pix.numel()
        state = (torch.ones(P, dtype=dtype), tmax > tmin, torch.zeros(P, 3, dtype=dtype), torch.zeros(P, dtype=dtype),
                 torch.zeros(P, dtype=dtype))
        for c0 in range(0, order.numel(), gauss_chunk):
            if not bool(state[1].any()):
                break
            sl = slice(c0, c0 + gauss_chunk)
            state = _composite_block(o, d, tmin, tmax, pos[sl], rows[sl], scl[sl], dns[sl], feat[sl], state)
        T, _, rgb, dist, hits = state
        outs.append(torch.cat([rgb, (1 - T)[:, None], dist[:, None], hits[:, None]], dim=1))
    out = torch.cat(outs, 0)
    return out[:, :4], out[:, 4], out[:, 5]


class _ReferenceUndoColour(torch.autograd.Function):
    """rgb = sum_k w_k max(colour_k, 0) with the BACKWARD the reference's sorted variant computes for it
    (gutKBufferRenderer.cuh:117-137 -> featuresIntegrateBwd, shRadiativeParticles.slang:179-207 around
    integrateRadiance<true>, :83-99): walking the hits front to back it un-does the back-to-front recurrence
    C_k = lerp(C_{k+1}, colour_k, alpha_k) from the forward's final colour,
        C_{k+1} = (C_k - colour_k alpha_k) / (1 - alpha_k),   d alpha_k += (colour_k - C_{k+1}) . G_k,
        d colour_k = alpha_k G_k,   G_{k+1} = (1 - alpha_k) G_k,   G_1 = dL/d rgb,
    but with the UNCLAMPED colour particleFeatures[idx] (the forward composited max(colour, 0), :159-161).  The colour
    gradient then passes the (colour > 0) mask of the SH backward (gaussianParticles.cuh:120-187)."""

    @staticmethod
    def forward(ctx, alpha, colour_unclamped):   # alpha [P,L] (0 where no hit), colour [P,L,3]
        one_m = 1 - alpha
        T_before = torch.cat([torch.ones_like(alpha[:, :1]), torch.cumprod(one_m, dim=1)[:, :-1]], dim=1)
        rgb = ((alpha * T_before)[..., None] * colour_unclamped.clamp(min=0)).sum(1)
        ctx.save_for_backward(alpha, colour_unclamped, rgb)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        alpha, col, rgb = ctx.saved_tensors
        P, L = alpha.shape
        C = rgb.clone()          # ray.featuresBackward: the forward's final colour
        G = g_rgb.clone()        # ray.featuresGradient
        d_alpha = torch.zeros_like(alpha)
        d_col = torch.zeros_like(col)
        for k in range(L):
            a = alpha[:, k]
            hit = a > 0
            wgt = torch.where(hit, 1.0 / (1.0 - a), torch.ones_like(a))
            Cn = torch.where(hit[:, None], (C - col[:, k] * a[:, None]) * wgt[:, None], C)
            d_alpha[:, k] = torch.where(hit, ((col[:, k] - Cn) * G).sum(-1), torch.zeros_like(a))
            d_col[:, k] = torch.where(hit[:, None], a[:, None] * G, torch.zeros_like(G)) * (col[:, k] > 0)
            G = torch.where(hit[:, None], (1.0 - a)[:, None] * G, G)
            C = Cn
        return d_alpha, d_col


def composite_ordered(params, tq, W, H, ray_ori, ray_dir, order_ids, order_count, sh_degree=3, dtype=torch.float64,
                      reference_undo_colour=False, kernel_degree=2):
    """Exact (autograd) compositing of each pixel's particles in a GIVEN order — the sorted variant's semantics
    (k_buffer_size > 0): the reference differentiates it with slang autodiff (gaussianParticles.slang:394-451,
    shRadiativeParticles.slang:179-207), i.e. true derivatives incl. min(0.99, .) and the hit distance.
    order_ids [P,L] (-1 padded) comes from the C oracle.  Returns rgba [P,4], dist [P].
    reference_undo_colour=True: same forward, but the colour's backward is the reference's own un-do recurrence with the
    unclamped colour (_ReferenceUndoColour) instead of the true derivative."""
    R, t, Rinv, cam_pos = pose_matrices(tq, dtype)
    pos, rot, scl, dns, sph = (params[k].to(dtype) for k in ("positions", "rotation", "scale", "density", "features"))
    feat_unclamped = precompute_features(pos, sph, cam_pos, sh_degree)
    feat_all = feat_unclamped.clamp(min=0)
    rows_all = quat_rows(rot)
    o, d = world_rays(torch.as_tensor(ray_ori, dtype=dtype), torch.as_tensor(ray_dir, dtype=dtype), Rinv, cam_pos)
    ids = torch.as_tensor(order_ids.astype("int64"))
    P, L = ids.shape
    valid = torch.arange(L)[None, :] < torch.as_tensor(order_count.astype("int64"))[:, None]
    idc = ids.clamp(min=0)
    mu, rws, s, sg, ft = pos[idc], rows_all[idc], scl[idc], dns[idc, 0], feat_all[idc]      # [P,L,...]
    gposc = o[:, None, :] - mu
    gro = torch.einsum("plij,plj->pli", rws, gposc) / s
    grdu = torch.einsum("plij,pj->pli", rws, d) / s
    grd = grdu / grdu.norm(dim=-1, keepdim=True)
    c = torch.cross(grd, gro, dim=-1)
    resp = kernel_response((c * c).sum(-1), kernel_degree)
    alpha = (resp * sg).clamp(max=ALPHA_MAX)
    alpha = torch.where(valid, alpha, torch.zeros_like(alpha))
    hit_t = (s * grd * (-(grd * gro).sum(-1))[..., None]).norm(dim=-1)
    one_m = 1 - alpha
    T_before = torch.cat([torch.ones(P, 1, dtype=dtype), torch.cumprod(one_m, dim=1)[:, :-1]], dim=1)
    w = alpha * T_before
    if reference_undo_colour:
        rgb = _ReferenceUndoColour.apply(alpha, feat_unclamped[idc])
    else:
        rgb = (w[..., None] * ft).sum(1)
    dist = (w * torch.where(valid, hit_t, torch.zeros_like(hit_t))).sum(1)
    T = one_m.prod(dim=1)
    return torch.cat([rgb, (1 - T)[:, None]], dim=1), dist
