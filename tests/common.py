"""Shared builders for the parity tests: seeded scenes + cameras in both forms (oracle dict / tracer Batch)."""
import importlib
import math

import numpy as np

cams = importlib.import_module("3dgrut_amd.cameras")
scenes = importlib.import_module("3dgrut_amd.scenes")
pose = importlib.import_module("3dgrut_amd.pose")


# OpenCV-fisheye polynomial of a ScanNet++ DSLR's size (k1..k4 of delta = theta (1 + k1 theta^2 + ... + k4 theta^8))
FISHEYE_DIST = dict(radial=[-0.03, -0.006, 0.001, -0.0003])


def make_view(kind, W, H, c2w, fx=None, fy=None, distortion=None):
    """Returns dict(W,H,c2w,ro,rd,oracle_cam,intrinsics_kw) for kind in {'pinhole','pinhole_list','fisheye'}."""
    tq = pose.sensor_pose_from_c2w(c2w).T_world_sensors[0]
    if kind in ("pinhole", "pinhole_list"):
        fx = fx or 1.0 * W
        fy = fy or fx
        ro, rd = cams.pinhole_rays(W, H, fx, fy)
        K = cams.pinhole_intrinsics_dict(W, H, fx, fy)
        if distortion:
            K["radial_coeffs"] = np.asarray(distortion["radial"], np.float32)
            K["tangential_coeffs"] = np.asarray(distortion["tangential"], np.float32)
            K["thin_prism_coeffs"] = np.asarray(distortion["thin_prism"], np.float32)
        ocam = dict(model="pinhole", principal_point=K["principal_point"], focal_length=K["focal_length"],
                    radial=K["radial_coeffs"], tangential=K["tangential_coeffs"], thin_prism=K["thin_prism_coeffs"],
                    pose_start=tq)
        if kind == "pinhole_list":
            # [fx,fy,cx,cy] path: focal -> fov -> focal round trip and orig_w=int(2cx) (tracer.py:386-403)
            w2, h2 = int(2 * (W / 2)), int(2 * (H / 2))
            fx2 = w2 / (2.0 * math.tan(0.5 * (2 * math.atan(w2 / (2 * fx)))))
            fy2 = h2 / (2.0 * math.tan(0.5 * (2 * math.atan(h2 / (2 * fy)))))
            ocam["focal_length"] = np.array([fx2, fy2], np.float32)
            ocam["principal_point"] = np.array([w2, h2], np.float32) / 2
            kw = dict(intrinsics=[fx, fy, W / 2, H / 2])
        else:
            kw = dict(intrinsics_OpenCVPinholeCameraModelParameters=K)
    elif kind == "fisheye":
        # distortion: dict(radial=(k1..k4)[, max_angle=...]) — the OpenCV-fisheye forward polynomial with non-zero coefficients
        # (cameraProjections.cuh:105-128) and, optionally, a field-of-view clamp tighter than the dataset rule's
        fx = fx or 0.45 * W
        fy = fy or fx
        radial = None if not distortion else distortion.get("radial")
        ro, rd = cams.fisheye_rays(W, H, fx, fy, radial=radial)
        K = cams.fisheye_intrinsics_dict(W, H, fx, fy, radial=radial, max_angle=None if not distortion else distortion.get("max_angle"))
        ocam = dict(model="fisheye", principal_point=K["principal_point"], focal_length=K["focal_length"],
                    radial=list(K["radial_coeffs"]), max_angle=K["max_angle"], pose_start=tq)
        kw = dict(intrinsics_OpenCVFisheyeCameraModelParameters=K)
    else:
        raise ValueError(kind)
    return dict(W=W, H=H, c2w=c2w, ro=ro, rd=rd, oracle_cam=ocam, intrinsics_kw=kw, tq=tq)


def to_batch(view, device):
    import torch
    gut = importlib.import_module("3dgrut_amd")
    return gut.Batch(rays_ori=torch.as_tensor(view["ro"], device=device), rays_dir=torch.as_tensor(view["rd"], device=device),
                     T_to_world=torch.as_tensor(view["c2w"], device=device)[None], **view["intrinsics_kw"])


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


# A colour difference above the fp32 tolerance is only acceptable where the oracle itself says a hit/no-hit (or early
# termination) decision along that ray sat within this many fp32-noise widths of its threshold (oracle_render_margins):
FLIP_MARGIN_BOUND = 6.0   # = K_BAND x 2.4; two CPU evaluations flip at margins up to 1.9 (derivation at the end of this file)
PIX_FLIP = 3.0            # x the pixel's own flip budget (as ROW_FLIP for gradient rows)
COLOUR_TOL = 2e-4         # flat colour tolerance of pixels without a flip-prone decision: two CPU evaluations differ by up to 8.7e-5 there


def check_colour_outliers(rgba_gpu, hits_gpu, ref, margins, tol=COLOUR_TOL, bound=FLIP_MARGIN_BOUND, label="", max_prone=0.05, dist_gpu=None,
                          budget=None):
    """Earns the "threshold flip" allowance instead of asserting it: every pixel whose colour (and, when given, hit distance)
    differs from the oracle by more than `tol`, or whose hit count differs, must be flip-prone (decision margin < bound); every
    pixel that is not flip-prone must be within `tol` and have the oracle's hit count.  Returns a small report (fractions, worst margin).

    budget ([H,W,2] from oracle.render_margins(..., budget_bound=ROW_FLIP_BOUND)): the QUANTITATIVE form, for frames on which a large
    share of the pixels has some decision near a threshold (rays that walk hundreds of faint entries): every pixel must then satisfy
    |difference| <= tol + PIX_FLIP x its own flip budget and |hit-count difference| <= its own count of flip-prone decisions, so that
    a flip-prone pixel is allowed what ITS near-threshold entries can move it by (2 alpha T each) and nothing more; `max_prone` and the
    flat 2.5e-2 cap do not apply."""
    H, W = ref["rgba"].shape[:2]
    diff = np.abs(np.asarray(rgba_gpu).reshape(H, W, 4) - ref["rgba"]).max(-1)
    if dist_gpu is not None:     # the integrated hit distance, relative to the frame's largest (it is a length, not a colour)
        diff = np.maximum(diff, np.abs(np.asarray(dist_gpu).reshape(H, W) - ref["dist"].reshape(H, W)) / max(1.0, float(np.abs(ref["dist"]).max())))
    hdiff = np.asarray(hits_gpu).reshape(H, W) != ref["hits"].reshape(H, W)
    m = margins.min(-1)
    prone = m < bound
    out = (diff > tol) | hdiff
    if budget is not None:
        allowed = tol + PIX_FLIP * budget[..., 0]
        hallowed = budget[..., 1]
        habs = np.abs(np.asarray(hits_gpu).reshape(H, W) - ref["hits"].reshape(H, W))
        over = (diff > allowed) | (habs > hallowed)
        rep = dict(outliers=int(out.sum()), flip_prone=int(prone.sum()), pixels=int(H * W), max_diff=float(diff.max()),
                   worst_vs_budget=float((diff / allowed).max()), pixels_with_budget=int((budget[..., 1] > 0).sum()),
                   worst_outlier_margin=float(m[out].max()) if out.any() else 0.0, over_budget=int(over.sum()))
        print(f"[outliers {label}] {rep}")
        assert not over.any(), f"{label}: {int(over.sum())} pixels differ by more than their own flip budget allows: {rep}"
        assert not (out & ~prone).any(), f"{label}: {(out & ~prone).sum()} pixels differ although no decision is near a threshold: {rep}"
        assert out.mean() <= 5e-3, rep
        return rep
    rep = dict(outliers=int(out.sum()), flip_prone=int(prone.sum()), pixels=int(H * W), max_diff=float(diff.max()),
               max_diff_not_prone=float(diff[~prone].max()) if (~prone).any() else 0.0,
               worst_outlier_margin=float(m[out].max()) if out.any() else 0.0)
    print(f"[outliers {label}] {rep}")
    assert not (out & ~prone).any(), f"{label}: {(out & ~prone).sum()} pixels differ although no decision is near a threshold: {rep}"
    assert diff.max() <= 2.5e-2, rep                  # a flipped hit moves a pixel by at most ~alpha*T*colour of that hit
    assert prone.mean() <= max_prone, rep             # the allowance stays a small part of the image
    assert out.mean() <= 2e-3, rep
    return rep


def check_tile_traversal(got_tiles, ref_tiles, margins, W, H, label="", bound=FLIP_MARGIN_BOUND):
    """Per-tile traversal depths (list entries fetched before every ray of the tile had ended; their sum is E of the byte model).
    A tile's depth is where its LAST ray ended, so it moves when one termination (or the hit that causes it) flips between two fp32
    evaluations: every tile whose depth differs from the oracle's must hold a pixel whose decision margin is below `bound` — the same
    evidence the colour check asks for.  Opaque scenes agree exactly; scenes whose rays barely saturate differ on a handful of tiles."""
    got = np.asarray(got_tiles).astype(np.int64).reshape(-1)
    ref = np.asarray(ref_tiles).astype(np.int64).reshape(-1)
    assert got.shape == ref.shape
    bad = np.nonzero(got != ref)[0]
    gx = (W + 15) // 16
    m = margins.min(-1)
    for t in bad:
        ty, tx = divmod(int(t), gx)
        tile_m = m[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
        assert tile_m.size and float(tile_m.min()) < bound, f"{label}: tile {t} walked {got[t]} entries, the oracle {ref[t]}, and no decision of its rays is near a threshold ({float(tile_m.min()):.1f})"
    assert bad.size <= max(2, 0.01 * got.size), f"{label}: {bad.size} of {got.size} tiles differ in traversal depth"
    print(f"[traversal {label}] {bad.size} of {got.size} tiles differ (all flip-prone), total {int(got.sum())} vs {int(ref.sum())}")
    return int(bad.size)


# ---- per-row gradient parity (a global relative L2 over a 6 M-row block hides a few thousand wrong rows) ----
GRAD_BLOCKS = (("positions", slice(0, 3)), ("density", slice(3, 4)), ("rotation", slice(4, 8)), ("scale", slice(8, 11)))
# `scale` of a block = the 99th-percentile row norm of the oracle's non-zero rows.  EVERY row must satisfy
#       |gpu - oracle| <= ROW_REL x |oracle row| + ROW_ABS x scale + ROW_NOISE x fp32_noise(row) + ROW_FLIP x flip_budget(row)
# where fp32_noise is the oracle's conditioning estimate of the row (sum over its hits of eps x nu x |contribution|: the
# response of a small, distant Gaussian carries a relative fp32 error of eps x nu, up to 1e-3, whatever the evaluation order),
# and flip_budget is what the ORACLE says a different but equally valid fp32 evaluation may move that row by, because a
# hit / no-hit decision within ROW_FLIP_BOUND noise widths of its threshold flips (oracle.backward(..., flip_bound=...);
# the reference's per-hit gradient is discontinuous there: a faint Gaussian's hit with alpha ~ 1/255 has d alpha / d sigma
# ~ 1).  Rows without a budget — no flip-prone decision on any ray touching them — get the fp32 terms only.  On top of that
# the rows above ROW_FLOOR x scale must have a relative error <= ROW_P999 at the 99.9th percentile, budget or not.
ROW_REL, ROW_ABS, ROW_NOISE, ROW_FLIP, ROW_FLOOR, ROW_P999 = 3e-3, 2e-4, 6.0, 3.0, 1e-3, 2e-2
ROW_FLIP_BOUND = 2.0 * FLIP_MARGIN_BOUND   # decisions within this many noise widths of a threshold count as flip-prone for the row budget


def check_gradient_rows(got, ref, label, budget=None, noise=None, rel_tol=ROW_REL, abs_tol=ROW_ABS, flip=ROW_FLIP, noise_k=ROW_NOISE,
                        floor=ROW_FLOOR, p999=ROW_P999, block_tol=2e-3):
    """Per-row comparison of one gradient block `got` [N,c] (GPU, fp32) with `ref` [N,c] (oracle, fp64 accumulation);
    `budget` [N]: the oracle's flip budget of the block (None: the distribution checks only).  Returns the report it prints."""
    import os
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    nr = np.linalg.norm(ref, axis=1)
    err = np.linalg.norm(got - ref, axis=1)
    nz = nr > 0
    scale = float(np.quantile(nr[nz], 0.99)) if nz.any() else 0.0
    big = nr > floor * scale
    rel = err[big] / nr[big] if big.any() else np.zeros(0)
    rep = dict(rows=int(nr.size), rows_nonzero=int(nz.sum()), rows_above_floor=int(big.sum()), scale=scale,
               block_rel_l2=rel_l2(got, ref),
               rel_p50=float(np.quantile(rel, 0.5)) if rel.size else 0.0, rel_p999=float(np.quantile(rel, 0.999)) if rel.size else 0.0,
               rel_max=float(rel.max()) if rel.size else 0.0, abs_max_over_scale=float(err.max() / scale) if scale > 0 else 0.0,
               gpu_nonzero_where_oracle_zero=int(((np.abs(got).max(1) > 0) & ~nz).sum()))
    if budget is not None:
        budget = np.asarray(budget, np.float64)
        tight = rel_tol * nr + abs_tol * scale + noise_k * (np.asarray(noise, np.float64) if noise is not None else 0.0)
        need = err > tight                                         # rows that need their flip allowance
        calm = ~need
        rep.update(block_rel_l2_without_rows_needing_budget=float(np.linalg.norm((got - ref)[calm]) / (np.linalg.norm(ref[calm]) + 1e-30)),
                   rows_with_budget=int((budget > 0).sum()), rows_needing_budget=int(need.sum()),
                   worst_row_vs_tight_bound_without_budget=float((err[budget == 0] / np.maximum(tight[budget == 0], 1e-300)).max()) if (budget == 0).any() else 0.0,
                   worst_row_vs_full_bound=float((err / np.maximum(tight + flip * budget, 1e-300)).max()),
                   rows_over_full_bound=int((err > tight + flip * budget).sum()))
    print(f"[rows {label}] {rep}")
    if os.environ.get("GUT_ROWS_REPORT_ONLY") == "1":
        return rep
    # block-wide relative L2: one flipped hit on one large row can carry the whole block's error (seen: a single row at 0.16 x scale
    # took a 300 k-row block to 2.3e-3), so with the per-row budget at hand the 2e-3 bar applies to the rows that did not need it
    # and the whole block gets 1e-2
    assert rep["block_rel_l2"] <= (block_tol if budget is None else 5 * block_tol), rep
    if budget is None:
        # distribution-only mode (no oracle budget at hand): the 99.9th-percentile relative error of the rows above the floor.  With a
        # budget every row has its own bound below, which is the statement; a fixed percentile is a property of the SCENE's conditioning
        # (0.3 - 1.7 % on the blob scenes, 3.5 % on the surface-like stand-in's flat discs, whose rows nevertheless sit at 0.35 of
        # their own bounds) and is reported, not asserted
        assert rep["rel_p999"] <= p999, rep
    if budget is not None:
        assert rep["block_rel_l2_without_rows_needing_budget"] <= block_tol, rep
        # no escape clause (round 4): the model bounds the measured two-evaluation band of the CPU oracle on every row with
        # K_BAND to spare (tests/test_cpu_oracle.py::test_two_fp32_evaluations_measure_the_tolerance_model); a row beyond it
        # means a missing term in the model or a wrong kernel
        assert rep["rows_over_full_bound"] == 0 and rep["worst_row_vs_full_bound"] <= 1.0, rep
        assert rep["rows_needing_budget"] <= 0.02 * max(1, rep["rows_nonzero"]), rep    # the allowance stays the exception
    return rep


def check_gradients_per_row(g12, g48, dens_g, sph_g, label, budget=None, sh_degree=3):
    """budget: [N,10] from oracle.backward(..., flip_bound=ROW_FLIP_BOUND) (positions, density, rotation, scale, colour).
    The SH row of a Gaussian is Y(dir) (x) dL/dRGB, and sum_k Y_k(dir)^2 = (degree + 1)^2 / (4 pi) for every direction, so the
    colour budget carries over to the [N,48] row with that factor."""
    for j, (name, sl) in enumerate(GRAD_BLOCKS):
        check_gradient_rows(np.asarray(g12)[:, sl], np.asarray(dens_g)[:, sl], f"{label}/{name}", None if budget is None else budget[:, j],
                            None if budget is None else budget[:, 5 + j])
    y = math.sqrt((sh_degree + 1) ** 2 / (4 * math.pi))
    check_gradient_rows(g48, sph_g, f"{label}/sh", None if budget is None else budget[:, 4] * y, None if budget is None else budget[:, 9] * y)


def check_side_stream_rows_are_gradient_free(owned_rows, g12, g48, dens_g, sph_g, label, min_rows=1):
    """Deviation 9 / the side-stream ownership argument as a per-row statement: on EVERY row of EVERY 64-row wave the side
    stream takes (waves without a tile, and waves the forward walked nothing of) the ORACLE's gradient — which terminates its
    rays by its own rule, independently of the GPU's per-tile depth bound — is exactly zero, and so is the GPU's."""
    owned = np.asarray(owned_rows, bool)
    assert int(owned.sum()) >= min_rows, (label, int(owned.sum()))
    for what, arr in (("oracle [N,12]", dens_g), ("oracle [N,48]", sph_g), ("gpu [N,12]", g12), ("gpu [N,48]", g48)):
        bad = np.abs(np.asarray(arr)[owned]).max(axis=1) > 0
        assert not bad.any(), f"{label}: {what} has {int(bad.sum())} non-zero rows among the {int(owned.sum())} rows of side-stream waves"
    print(f"[side-stream rows {label}] {int(owned.sum())} rows in side-stream waves, oracle and GPU gradients exactly zero on all of them")


def fisheye_max_angle_edge_case(ulps_above=0):
    """SURVEY §8c's known-answer case "fisheye theta == max_angle": a distorted OpenCV-fisheye camera at the world origin with the
    identity pose (camera space = world space, exactly) whose max_angle is the fp32 value atan2f(rho, z) of Gaussian 0's centre
    (plus `ulps_above` ulps).  Gaussian 0 is opaque and degenerate (scale 1e-12: all seven sigma points round to its centre), so it
    lives or dies by the comparison `theta < max_angle` alone (cameraProjections.cuh:119,127: theta = min(thetaFull, maxAngle),
    valid iff theta < maxAngle): with ulps_above = 0 every sigma point is invalid and the Gaussian gets no tile, one ulp above it
    is valid and gets one.  The other Gaussians straddle the cone.  Returns (scene, view, theta_star)."""
    import importlib
    oracle = importlib.import_module("oracle.oracle")
    sc = scenes.scene_c1(400, 21)
    sc["positions"][:, 2] += np.float32(2.2)
    x, y, z = np.float32(0.4), np.float32(0.3), np.float32(1.0)
    sc["positions"][0] = (x, y, z)
    sc["scale"][0] = 1e-12
    sc["density"][0] = 0.999
    rho = np.sqrt(np.float32(x * x) + np.float32(y * y), dtype=np.float32)
    theta = np.float32(oracle.lib().oracle_det_atan2f(float(rho), float(z)))
    for _ in range(ulps_above):
        theta = np.nextafter(theta, np.float32(4.0))
    W, H = 144, 96
    view = make_view("fisheye", W, H, np.eye(4, dtype=np.float32), distortion=dict(FISHEYE_DIST, max_angle=float(theta)))
    assert np.float32(view["oracle_cam"]["max_angle"]) == theta
    return sc, view, float(theta)


def densified_like_scene(n=20000, seed=5):
    """A CPU-sized stand-in for what clone / split / training leave behind (tests/test_gpu_densify.py trains a real one on the GPU):
    a lego-like scene with 30 % of the Gaussians cloned in place, 20 % split into two children drawn inside the parent (scales / 1.6,
    gs.py:117-160) and opacities pushed up — coincident and nested Gaussians, many opaque rays whose deep entries carry tiny
    gradients.  Used by the two-evaluation tolerance experiment of tests/test_cpu_oracle.py."""
    rng = np.random.default_rng(seed)
    sc = scenes.scene_lego_like(n, seed)
    clone = rng.random(n) < 0.3
    split = (~clone) & (rng.random(n) < 0.25)
    parts = [{k: v[~split] for k, v in sc.items()}, {k: v[clone] for k, v in sc.items()}]
    par = {k: v[split] for k, v in sc.items()}
    q = par["rotation"].astype(np.float64)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    for _ in range(2):
        child = {k: v.copy() for k, v in par.items()}
        off = np.einsum("nij,nj->ni", R, par["scale"] * rng.normal(size=par["scale"].shape))
        child["positions"] = (par["positions"] + off).astype(np.float32)
        child["scale"] = (par["scale"] / 1.6).astype(np.float32)
        parts.append(child)
    out = {k: np.concatenate([p[k] for p in parts]) for k in sc}
    logit = np.log(out["density"] / (1 - out["density"])) + 2.0
    out["density"] = (1 / (1 + np.exp(-logit))).astype(np.float32).clip(0, 0.999)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# The tolerance model's constants and where they come from (round 4; they do not move: a failing GPU test means a term is missing
# from the model or a kernel is wrong, not that a constant is too small).
#
# oracle/gut_oracle.c evaluates the per-(ray, Gaussian) formulas of the compositing passes in THREE equally valid fp32 forms:
# variant 0 (the reference's operation order, no contraction, libm expf), variant 1 (what the HIP kernels do: rows of diag(1/s) R^T
# rounded once, FMAs, |u x o|^2 / |u|^2 with one reciprocal, exp2 of the pre-scaled argument) and variant 2 (variant 1 with the
# reciprocal / reciprocal-square-root / exp2 results moved by -1 / 0 / +1 ulp, the hardware instructions' specified accuracy).
# tests/test_cpu_oracle.py::test_two_fp32_evaluations_measure_the_tolerance_model MEASURES |variant 0 - variant v| per pixel and
# per gradient row on six scenes (toy pinhole, distorted fisheye, big dense splats, 60 k lego-like at 400 x 400, 33 k
# densified-like, 400 k flat discs of the surface-like stand-in) and asserts that the model below, with every constant divided by
# K_BAND, bounds that band on EVERY pixel and EVERY row, no exceptions:
#     measured                                                     asserted on the CPU           GPU bound (x K_BAND)
#     largest decision margin at which two evaluations flip: 1.43  <= FLIP_MARGIN_BOUND / K_BAND  FLIP_MARGIN_BOUND = 6
#     largest |row difference| / noise(row), rows without budget:  <= ROW_NOISE / K_BAND          ROW_NOISE = 6
#       2.2 (rotation rows of flat discs), 1.0 elsewhere
#     largest |row difference| / (full bound), all rows: 0.30      <= 1 / K_BAND                  ROW_FLIP = 3, ROW_REL, ROW_ABS
#     largest colour difference of a pixel without a flip-prone    <= COLOUR_TOL / 2              COLOUR_TOL = 2e-4
#       decision: 8.7e-5
#     largest |pixel difference| / (COLOUR_TOL + PIX_FLIP x the     <= 1 / 2                       PIX_FLIP = 3
#       pixel's own flip budget): 0.43 (that calm pixel), 0.35 on pixels with a budget
# The noise model itself was completed with this experiment (gut_oracle.c: render_bwd_impl): it showed rows of two CPU evaluations
# apart by 73 x the round-3 estimate — deep entries of opaque rays, whose G inherits an ABSOLUTE error from the residual form
# (T_final = 1 - alpha_out, (rgb_final - rgb_run) / T'), the noise of every alpha in front of them (transmittance chain), and, on
# rays whose termination is flip-prone, the other ending's finals.  Those are the three terms added in round 4; a fourth followed when
# the GPU met the surface-like stand-in (two hit-count flips at 6.4 and 11.5 "noise widths"): the response's noise estimate assumed
# isotropic Gaussians, and for flat 8 : 1 discs the canonical-space direction and origin carry |1/s|_max / |1/s|_effective times
# more rounding error (hit_noise in gut_oracle.c) — with it those two decisions sit at 1 - 2 widths, and the CPU figures above are
# the ones measured with it.
# K_BAND = 2.5 is the one factor between "two CPU evaluations" and "the GPU": the GPU adds fp32 (instead of double) sums over
# the pixels of a wave and float atomics across waves on top of what variant 2 models.
K_BAND = 2.5
