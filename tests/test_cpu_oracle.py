"""CPU tests of the oracle: known-answer tests, structural invariants, C oracle vs the independent float64
PyTorch autograd restatement, and the host pose math against the golden vectors generated from the
reference's own Python (tests/golden/gen_pose_golden.py)."""
import importlib
import math
import os

import numpy as np
import pytest
import torch

from tests.common import (COLOUR_TOL, FISHEYE_DIST, FLIP_MARGIN_BOUND, GRAD_BLOCKS, K_BAND, PIX_FLIP, ROW_ABS, ROW_FLIP, ROW_FLIP_BOUND, ROW_NOISE, ROW_REL, cams,
                          densified_like_scene, fisheye_max_angle_edge_case, make_view, pose, scenes)

oracle = importlib.import_module("oracle.oracle")
prt = importlib.import_module("oracle.per_ray_torch")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _single(pos, scale=0.1, dens=0.8, rgb0=(0.3, -0.2, 1.0)):
    sc = dict(positions=np.array([pos], np.float32), rotation=np.array([[1, 0, 0, 0]], np.float32),
              scale=np.full((1, 3), scale, np.float32), density=np.array([[dens]], np.float32),
              features=np.zeros((1, 48), np.float32))
    sc["features"][0, :3] = rgb0
    return sc


def test_pose_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "pose_golden.npz"))
    for c2w, tq in zip(g["c2w"], g["tquat"]):
        p = pose.sensor_pose_from_c2w(c2w)
        assert p.timestamps_us == [0, 1]
        assert np.abs(p.T_world_sensors[0] - tq).max() <= 1e-6
        assert np.array_equal(p.T_world_sensors[0], p.T_world_sensors[1])


def test_single_gaussian_on_axis_closed_form():
    """alpha = min(.99, sigma*exp(-d^2/2)), rgb = max(0.28209479*c0+0.5, 0)*alpha, hitT = distance."""
    W = H = 33
    view = make_view("pinhole", W, H, np.eye(4, dtype=np.float32), fx=40.0)
    sc = _single((0.0, 0.0, 3.0))
    out = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"], sh_degree=0)
    assert out["tiles_count"][0] >= 1 and out["visibility"][0] == 1
    c = out["rgba"][16, 16]
    col = np.maximum(0.28209479177387814 * np.array([0.3, -0.2, 1.0]) + 0.5, 0)
    assert abs(c[3] - 0.8) < 1e-6 and np.abs(c[:3] - col * 0.8).max() < 1e-6
    assert abs(out["dist"][16, 16, 0] - 0.8 * 3.0) < 1e-5 and out["hits"][16, 16, 0] == 1
    # off-axis pixel: response follows exp(-d^2/2) with d = perpendicular distance / scale
    d = view["rd"][0, 16, 20]
    perp = np.linalg.norm(np.cross(d, np.array([0, 0, 3.0]))) / 0.1
    a = 0.8 * math.exp(-0.5 * perp * perp)
    exp_a = a if (math.exp(-0.5 * perp * perp) > 0.0113 and a > 1 / 255) else 0.0
    assert abs(out["rgba"][16, 20, 3] - exp_a) < 1e-5


def test_front_to_back_order_and_saturation():
    W = H = 17
    view = make_view("pinhole", W, H, np.eye(4, dtype=np.float32), fx=30.0)
    a, b = _single((0, 0, 2.0), dens=0.6, rgb0=(1.5, 1.5, 1.5)), _single((0, 0, 4.0), dens=0.9, rgb0=(-1.0, -1.0, -1.0))
    sc = {k: np.concatenate([b[k], a[k]]) for k in a}  # far one first in memory: order must come from depth keys
    out = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"], sh_degree=0)
    k = out["sorted_ids"][(out["sorted_keys"] >> np.uint64(32)) == 0]
    assert list(k[:2]) == [1, 0]
    ca = max(0.28209479177387814 * 1.5 + 0.5, 0); cb = max(0.28209479177387814 * -1.0 + 0.5, 0)
    px = out["rgba"][8, 8]
    assert abs(px[0] - (0.6 * ca + 0.4 * 0.9 * cb)) < 1e-6 and abs(px[3] - (1 - 0.4 * 0.1)) < 1e-6
    # density 1.0 saturates at alpha 0.99
    s = _single((0, 0, 2.0), dens=1.0)
    o2 = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(s), s["features"], view["ro"], view["rd"], sh_degree=0)
    assert abs(o2["rgba"][8, 8, 3] - 0.99) < 1e-6


def test_culling_rules():
    W = H = 32
    view = make_view("pinhole", W, H, np.eye(4, dtype=np.float32), fx=30.0)
    cases = {"low_opacity": _single((0, 0, 3.0), dens=1 / 255 - 1e-5), "behind_near": _single((0, 0, 0.19)),
             "outside_margin": _single((30.0, 0, 3.0), scale=0.01), "ok": _single((0, 0, 3.0), dens=0.05)}
    res = {k: oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(v), v["features"], view["ro"], view["rd"])["tiles_count"][0]
           for k, v in cases.items()}
    assert res["low_opacity"] == 0 and res["behind_near"] == 0 and res["outside_margin"] == 0 and res["ok"] >= 1


def test_fisheye_theta_equal_to_max_angle_is_invalid_one_ulp_above_is_valid():
    """SURVEY §8c KAT: OpenCV fisheye, theta == max_angle (cameraProjections.cuh:119,127: theta = min(thetaFull, maxAngle), the
    projection is valid iff theta < maxAngle).  A degenerate opaque Gaussian whose seven sigma points all sit at exactly
    theta == max_angle gets no tile; with max_angle one fp32 ulp larger it is projected (non-zero polynomial coefficients) and
    gets its tile at the closed-form position f * p.xy * theta (1 + k1 t^2 + k2 t^4 + k3 t^6 + k4 t^8) / rho + pp."""
    out = {}
    for ulps in (0, 1):
        sc, view, theta = fisheye_max_angle_edge_case(ulps)
        out[ulps] = oracle.forward(view["oracle_cam"], 144, 96, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    assert out[0]["tiles_count"][0] == 0 and out[0]["visibility"][0] == 0
    assert out[1]["tiles_count"][0] == 1 and out[1]["visibility"][0] == 1
    k1, k2, k3, k4 = FISHEYE_DIST["radial"]
    t2 = theta * theta
    delta = theta * (1.0 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) / 0.5
    fx = 0.45 * 144
    assert np.abs(out[1]["proj_pos"][0] - (fx * 0.4 * delta + 72.0, fx * 0.3 * delta + 48.0)).max() <= 2e-4
    # the cone cuts through the other Gaussians too: some lose tiles against the dataset rule's max_angle, none gains one
    sc, view, _ = fisheye_max_angle_edge_case(0)
    wide = dict(view["oracle_cam"], max_angle=1.3)
    ref_wide = oracle.forward(wide, 144, 96, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    assert ref_wide["M"] > out[0]["M"] > 100


def test_structural_invariants_and_pad_aliasing():
    sc = scenes.scene_c1(600, 13)
    W, H = 100, 70
    view = make_view("pinhole", W, H, cams.look_at_c2w((0.2, 0.1, -3.0), (0, 0, 0)), fx=80.0)
    o = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    T = 7 * 5
    assert o["end_bit"] == 32 + 6 and oracle.higher_msb(T) == 6
    for n in (1, 2, 3, 63, 64, 65, 2500, 4056, 65535, 65536):
        assert oracle.higher_msb(n) == n.bit_length()
    assert np.array_equal(np.cumsum(o["tiles_count"], dtype=np.uint32), o["tiles_offset"]) and o["M"] == int(o["tiles_count"].sum())
    mask = (np.uint64(1) << np.uint64(o["end_bit"])) - np.uint64(1)
    mk = o["sorted_keys"] & mask
    assert np.all(mk[1:] >= mk[:-1])
    # stability: equal keys keep emission (particle-index) order
    same = mk[1:] == mk[:-1]
    assert np.all(o["sorted_ids"][1:][same] > o["sorted_ids"][:-1][same])
    tiles = (o["sorted_keys"] >> np.uint64(32)).astype(np.int64)
    for t in range(T):
        b, e = o["tile_ranges"][t]
        assert np.all(tiles[b:e] == t) and (e - b) == int((tiles == t).sum())
    # invalid keys (tile 0xFFFFFFFF) alias to 2^bw-1 >= T under the truncated sort and therefore sort last
    keys = np.array([(3 << 32) | 5, (0xFFFFFFFF << 32) | 0x7F7FFFFF, (1 << 32) | 9], np.uint64)
    ids = np.array([10, 0xFFFFFFFF, 11], np.uint32)
    ko = np.zeros(3, np.uint64); io = np.zeros(3, np.uint32)
    import ctypes as C
    oracle.lib().oracle_sort_pairs(C.c_uint32(3), C.c_int(32 + 6), keys.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                   ko.ctypes.data_as(C.c_void_p), io.ctypes.data_as(C.c_void_p))
    assert list(io) == [11, 10, 0xFFFFFFFF]
    rng = np.zeros((T, 2), np.uint32)
    oracle.lib().oracle_tile_ranges(C.c_uint32(3), C.c_uint32(T), ko.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p))
    assert list(rng[1]) == [0, 1] and list(rng[3]) == [1, 2]


def test_deterministic_log_and_atan():
    L = oracle.lib()
    xs = np.concatenate([np.geomspace(1e-30, 1e30, 400), np.linspace(0.5, 2.0, 300), [1.0, 255.0]]).astype(np.float32)
    got = np.array([L.oracle_det_logf(float(x)) for x in xs])
    assert np.abs(got - np.log(xs.astype(np.float64))).max() <= 3e-7 * np.maximum(1.0, np.abs(np.log(xs.astype(np.float64)))).max()
    ys = np.geomspace(1e-6, 10, 60).astype(np.float32)
    for y in ys:
        for x in (-5.0, -0.3, 0.0, 0.2, 1.0, 7.0):
            assert abs(L.oracle_det_atan2f(float(y), float(x)) - math.atan2(y, x)) <= 4e-7


@pytest.mark.parametrize("kind", ["pinhole", "fisheye"])
def test_c_oracle_matches_float64_autograd(kind):
    sc = scenes.scene_c1(250, seed=5)
    W, H = 48, 40
    view = make_view(kind, W, H, cams.look_at_c2w((0.3, 0.2, -3.5 if kind == "pinhole" else -2.0), (0, 0, 0)), fx=50 if kind == "pinhole" else None)
    f = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sc.items()}
    rgba, dist, hits = prt.render_tiled(params, view["tq"], W, H, view["ro"], view["rd"], f["tile_ranges"], f["sorted_ids"])
    assert np.abs(rgba.detach().numpy() - f["rgba"]).max() <= 2e-5
    assert np.abs(hits.numpy() - f["hits"][..., 0]).max() == 0
    rg = np.random.default_rng(0).normal(size=(H, W, 4)).astype(np.float32)
    (rgba * torch.tensor(rg, dtype=torch.float64)).sum().backward()
    dg, sg, _ = oracle.backward(view["oracle_cam"], f, rg, np.zeros((H, W, 1), np.float32))
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    assert rel(params["positions"].grad.numpy(), dg[:, 0:3]) <= 5e-5
    assert rel(params["density"].grad.numpy()[:, 0], dg[:, 3]) <= 5e-5
    assert rel(params["rotation"].grad.numpy(), dg[:, 4:8]) <= 5e-5
    assert rel(params["scale"].grad.numpy(), dg[:, 8:11]) <= 5e-5
    assert rel(params["features"].grad.numpy(), sg) <= 5e-5
    # projection (K1) against the vectorised torch restatement
    pr = prt.project(view["oracle_cam"], view["tq"], W, H, params)
    assert np.array_equal(pr["valid"].numpy(), f["visibility"] > 0)
    cnt = f["tiles_count"] > 0
    assert np.abs(pr["center"].numpy()[cnt] - f["proj_pos"][cnt]).max() <= 5e-4
    assert np.abs(pr["extent"].numpy()[cnt] - f["extent"][cnt]).max() <= 5e-4


def test_golden_c1_fixture():
    """Committed fixture of BASELINE configs[0] (1k Gaussians, 128x128): guards the oracle itself against drift."""
    g = np.load(os.path.join(GOLD, "c1_golden.npz"))
    sc = scenes.scene_c1(1000, 0)
    view = make_view("pinhole", 128, 128, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=128.0)
    o = oracle.forward(view["oracle_cam"], 128, 128, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    assert o["M"] == int(g["M"]) and np.array_equal(o["tiles_count"], g["tiles_count"])
    assert np.array_equal(o["sorted_ids"], g["sorted_ids"]) and np.array_equal(o["sorted_keys"], g["sorted_keys"])
    assert np.array_equal(o["tile_ranges"], g["tile_ranges"])
    assert np.abs(o["rgba"][::4, ::4] - g["rgba_sub"]).max() <= 1e-6
    dg, sg, _ = oracle.backward(view["oracle_cam"], o, g["rgba_grad"], np.zeros((128, 128, 1), np.float32))
    assert np.abs(dg.sum(0) - g["density_grad_colsum"]).max() <= 1e-6 * max(1.0, np.abs(g["density_grad_colsum"]).max())


@pytest.mark.parametrize("K", [1, 3, 16])
def test_kbuffer_oracle_matches_ordered_autograd(K):
    """oracle_render_kbuffer (sorted variant) vs float64 compositing of the order it recorded; K=1 == unsorted."""
    sc = scenes.scene_c1(300, seed=8)
    W, H = 40, 32
    view = make_view("pinhole", W, H, cams.look_at_c2w((0.2, -0.1, -3.0), (0, 0, 0)), fx=44)
    f = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    kb = oracle.render_kbuffer(view["oracle_cam"], f, K=K, max_order=int(f["hits"].max()) + 32)
    assert kb["order_count"].max() > 0
    if K == 1:
        assert np.abs(kb["rgba"] - f["rgba"]).max() == 0 and np.abs(kb["dist"] - f["dist"]).max() == 0
    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in sc.items()}
    rgba, dist = prt.composite_ordered(params, view["tq"], W, H, view["ro"], view["rd"], kb["order_ids"], kb["order_count"])
    assert np.abs(rgba.numpy().reshape(H, W, 4) - kb["rgba"]).max() <= 3e-5
    assert np.abs(dist.numpy().reshape(H, W, 1) - kb["dist"]).max() <= 3e-5 * max(1.0, float(kb["dist"].max()))
    # hit distances composited in non-decreasing order once K covers the local overlap
    assert np.array_equal(kb["hits"][..., 0].reshape(-1), kb["order_count"].astype(np.float32))


def test_decision_margins_flag_a_response_sitting_on_the_threshold():
    """oracle_render_margins: one isotropic Gaussian on the optical axis whose density makes resp*sigma cross 1/255 at a known
    pixel radius: pixels on that ring are flip-prone (margin < 4), pixels well inside or outside are not."""
    W = H = 64
    view = make_view("pinhole", W, H, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=64.0)
    d12 = np.zeros((1, 12), np.float32)
    d12[0, 3] = 0.5
    d12[0, 4] = 1.0
    d12[0, 8:11] = 0.3
    sph = np.zeros((1, 48), np.float32)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"])
    m = oracle.render_margins(view["oracle_cam"], ref)
    assert m.shape == (H, W, 2) and ref["M"] > 0
    hit = ref["hits"].reshape(H, W) > 0
    assert hit.any() and (~hit).any()
    # the boundary of the hit region is where the margin collapses; the centre and the far corner are calm
    thr = m[..., 0]
    edge = hit & ~(np.roll(hit, 1, 0) & np.roll(hit, -1, 0) & np.roll(hit, 1, 1) & np.roll(hit, -1, 1))
    assert thr[H // 2, W // 2] > 1e4 and np.median(thr[edge]) < np.median(thr[hit & ~edge])
    assert (thr[hit] < 4.0).mean() < 0.05


def test_reference_undo_colour_backward_equals_the_true_derivative_without_negative_colours():
    """oracle/per_ray_torch._ReferenceUndoColour (the reference's sorted-variant colour backward, un-do form): with all colours
    >= 0 it is the exact derivative of the compositing sum; with a negative channel its alpha gradient is not."""
    import torch
    prt = importlib.import_module("oracle.per_ray_torch")
    g = torch.Generator().manual_seed(0)
    P, L = 50, 7
    up = torch.randn(P, 3, generator=g, dtype=torch.float64)

    def both(col0):
        out = []
        for ref in (True, False):
            alpha = (torch.rand(P, L, generator=torch.Generator().manual_seed(1), dtype=torch.float64) * 0.9).requires_grad_(True)
            col = col0.clone().requires_grad_(True)
            if ref:
                rgb = prt._ReferenceUndoColour.apply(alpha, col)
            else:
                Tb = torch.cat([torch.ones(P, 1, dtype=torch.float64), torch.cumprod(1 - alpha, 1)[:, :-1]], 1)
                rgb = ((alpha * Tb)[..., None] * col.clamp(min=0)).sum(1)
            (rgb * up).sum().backward()
            out.append((rgb.detach(), alpha.grad, col.grad))
        return out
    pos = torch.rand(P, L, 3, generator=g, dtype=torch.float64)
    (r1, a1, c1), (r2, a2, c2) = both(pos)
    assert float((r1 - r2).abs().max()) == 0 and float((a1 - a2).abs().max()) < 1e-13 and float((c1 - c2).abs().max()) < 1e-13
    neg = pos.clone(); neg[:, 2, 0] = -0.7
    (r1, a1, c1), (r2, a2, c2) = both(neg)
    assert float((r1 - r2).abs().max()) == 0 and float((c1 - c2).abs().max()) < 1e-13
    assert float((a1 - a2).abs().max()) > 1e-3


def test_flip_budget_covers_what_a_threshold_perturbation_moves():
    """oracle.backward(..., flip_bound=b) exports, per Gaussian, how far a flipped hit / no-hit decision may move each gradient
    row (the allowance tests/common.check_gradient_rows grants the GPU per row).  Pinned here against an independent
    experiment: the same backward with min_response and min_alpha moved by 1e-4 relative — every decision that flips under that
    perturbation sits within 1e-4 / (eps * nu) <= 839 noise widths (nu >= 2), i.e. is flip-prone for b = 1000 — must stay within
    tight fp32 terms + the budget on EVERY row, must actually move some rows (else the test is vacuous), and rows without a
    budget must not move at all.  Also: asking for the budget does not change the gradients."""
    from tests.common import GRAD_BLOCKS
    sc = scenes.scene_c1(3000, 13)
    sc["density"][::2] *= 0.02          # faint Gaussians: hits near the min_alpha threshold, where d alpha / d sigma ~ 1
    W, H = 160, 128
    view = make_view("pinhole", W, H, cams.look_at_c2w((0.2, 0.1, -3.0), (0, 0, 0)), fx=150)
    d12 = scenes.pack_density(sc)
    fwd = oracle.forward(view["oracle_cam"], W, H, d12, sc["features"], view["ro"], view["rd"])
    rg = np.random.default_rng(3).normal(size=(H, W, 4)).astype(np.float32)
    dg = np.zeros((H, W, 1), np.float32)
    base = oracle.backward(view["oracle_cam"], fwd, rg, dg)
    with_b = oracle.backward(view["oracle_cam"], fwd, rg, dg, flip_bound=1000.0)
    assert all(np.allclose(base[j], with_b[j], rtol=1e-12, atol=1e-300) for j in range(3)) and np.abs(with_b[1]).max() > 0
    budget = with_b[3]
    assert budget.shape == (3000, 10) and (budget >= 0).all() and 0 < (budget[:, 1] > 0).sum() < 3000
    none = oracle.backward(view["oracle_cam"], fwd, rg, dg, flip_bound=0.0)[3]
    assert not none[:, :5].any() and np.allclose(none[:, 5:], budget[:, 5:], rtol=1e-12)
    moved_rows = 0
    for sign in (+1.0, -1.0):
        prm = oracle.default_params()
        prm.min_kernel_density *= (1.0 + sign * 1e-4)
        prm.alpha_threshold *= (1.0 + sign * 1e-4)
        fwd_p = oracle.forward(view["oracle_cam"], W, H, d12, sc["features"], view["ro"], view["rd"], params=prm)
        pert = oracle.backward(view["oracle_cam"], fwd_p, rg, dg, params=prm)
        blocks = [(base[0][:, sl], pert[0][:, sl]) for _, sl in GRAD_BLOCKS] + [(base[2], pert[2])]
        for j, (a, b) in enumerate(blocks):
            err = np.linalg.norm(a - b, axis=1)
            nr = np.linalg.norm(a, axis=1)
            scale = np.quantile(nr[nr > 0], 0.99)
            bound = 1e-9 * nr + 1e-12 * scale + 2.0 * budget[:, j]
            assert (err <= bound).all(), (sign, j, float((err / np.maximum(bound, 1e-300)).max()))
            assert not err[budget[:, j] == 0].any()
            moved_rows += int((err > 1e-9 * nr + 1e-12 * scale).sum())
    assert moved_rows > 0


@pytest.mark.parametrize("kind,W,H", [("pinhole", 128, 96), ("fisheye", 144, 96), ("pinhole", 100, 70)])
def test_tile_rules_against_an_independent_float64_restatement(kind, W, H):
    """K1's tile bounding box and per-tile culling (gutProjector.cuh:32-78) — the rules every integer buffer rests on, and which
    the GPU shares with the C oracle bit for bit — against a second, vectorised float64 restatement (per_ray_torch.tile_footprints)
    evaluated densely for every (Gaussian, tile) pair:
      * the set of tiles each Gaussian keeps is identical, except pairs whose power sits within 1e-4 (relative) of the threshold
        or whose box edge sits within 1e-3 px of a tile boundary (fp32 vs fp64);
      * soundness: the reference evaluates the footprint's quadratic form at ONE point of the tile, so its value bounds the true
        minimum over the tile from above — every kept tile provably intersects the footprint;
      * how much the one-point rule gives away (tiles culled although the true minimum is below the threshold) is reported."""
    sc = scenes.scene_c1(1500, seed=29)
    sc["scale"] *= np.random.default_rng(1).uniform(0.3, 3.0, size=(1500, 1)).astype(np.float32)   # sub-tile splats up to multi-tile ones
    eye = (0.3, 0.2, -3.5) if kind == "pinhole" else (0.2, 0.1, -1.6)
    view = make_view(kind, W, H, cams.look_at_c2w(eye, (0, 0, 0)), fx=0.9 * W if kind == "pinhole" else None)
    f = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in sc.items()}
    pr = prt.project(view["oracle_cam"], view["tq"], W, H, params)
    fp = prt.tile_footprints(pr, W, H)
    gx, gy = fp["grid"]
    T = gx * gy
    # the oracle's kept pairs, from its unsorted key list (tile index in the high word), as a dense [N,T] matrix
    kept_c = np.zeros((1500, T), bool)
    kept_c[f["unsorted_ids"].astype(np.int64), (f["unsorted_keys"] >> np.uint64(32)).astype(np.int64)] = True
    assert kept_c.sum() == f["M"] and np.array_equal(kept_c.sum(1), f["tiles_count"])
    kept_t = fp["kept"].numpy()
    power, thr = fp["power"].numpy(), fp["threshold"].numpy()[:, None]
    c, e = pr["center"].numpy(), pr["extent"].numpy()
    edge = np.minimum.reduce([np.abs(((c[:, 0] - 0.5 - e[:, 0]) / 16 + 0.5) % 1 - 0.5), np.abs(((c[:, 0] - 0.5 + e[:, 0]) / 16 + 0.5) % 1 - 0.5),
                              np.abs(((c[:, 1] - 0.5 - e[:, 1]) / 16 + 0.5) % 1 - 0.5), np.abs(((c[:, 1] - 0.5 + e[:, 1]) / 16 + 0.5) % 1 - 0.5)])
    touchy = (np.abs(power - thr) <= 1e-4 * np.maximum(np.abs(thr), 1.0)) | (edge[:, None] < 1e-3 / 16)
    differ = kept_c != kept_t
    assert not (differ & ~touchy).any(), f"{int((differ & ~touchy).sum())} (Gaussian, tile) pairs differ away from any threshold"
    assert differ.sum() <= 5 and kept_t.sum() > 2000 and (fp["in_box"].numpy() & ~kept_t).sum() > 200   # culling is exercised both ways
    # soundness of the one-point rule
    exact = fp["exact_min_power"].numpy()
    assert (exact <= power + 1e-9 * np.maximum(np.abs(power), 1.0)).all()
    assert (exact[kept_t] < thr.repeat(T, 1)[kept_t] * (1 + 1e-9) + 1e-12).all()
    lost = fp["in_box"].numpy() & ~kept_t & (exact < thr)
    print(f"[tile rule {kind} {W}x{H}] kept {int(kept_t.sum())} of {int(fp['in_box'].numpy().sum())} boxed pairs; "
          f"{int(lost.sum())} culled although the footprint reaches the tile (the reference's one-point rule)")


def _big_splats():
    sc = scenes.scene_c1(400, 7)
    sc["scale"] = (sc["scale"] * 4.0).astype(np.float32)
    sc["density"] = np.clip(sc["density"] * 1.5, 0, 0.999).astype(np.float32)
    return sc


BAND_SCENES = {
    "c1_pinhole_128": (lambda: scenes.scene_c1(1000, 0), "pinhole", 128, 128, lambda: cams.look_at_c2w((0, 0, -4), (0, 0, 0)), dict(fx=128)),
    "fisheye_distorted": (lambda: scenes.scene_c1(900, 12), "fisheye", 144, 96, lambda: cams.look_at_c2w((0.1, 0.0, -1.5), (0, 0, 0.5)),
                          dict(distortion=FISHEYE_DIST)),
    "dense_big_splats": (_big_splats, "pinhole", 64, 64, lambda: cams.look_at_c2w((0, 0, -3), (0, 0, 0)), dict(fx=64)),
    "lego_like_60k_400": (lambda: scenes.scene_lego_like(60000, 1), "pinhole_list", 400, 400, lambda: cams.orbit_c2w(4.0, 30, 20), dict(fx=555.0)),
    "densified_like_30k_320": (lambda: densified_like_scene(20000, 5), "pinhole", 320, 320, lambda: cams.orbit_c2w(3.6, -40, 25), dict(fx=420.0)),
    # flat 8 : 1 discs on shells, mostly faint, seen at grazing angles and from far away (the bench's surface-like stand-in, thinned):
    # the anisotropy terms of the noise estimate (gut_oracle.c: hit_noise) are what this scene measures
    "surface_like_400k_384x256": (lambda: scenes.scene_surface_like(400_000, 6, scale_mu=math.log(0.05)), "pinhole", 384, 256,
                                  lambda: cams.orbit_c2w(4.5, 7.0, 12.0), dict(fx=320.0)),
}


@pytest.mark.parametrize("name", list(BAND_SCENES))
def test_two_fp32_evaluations_measure_the_tolerance_model(name):
    """The measured basis of the GPU tolerances (tests/common.py, "The tolerance model's constants"): the oracle's compositing
    passes evaluated in a second and a third equally valid fp32 form (oracle.variant(1): the HIP kernels' own form — FMAs,
    pre-scaled rows, one reciprocal, exp2; variant(2): the same with 1-ulp reciprocal / rsqrt / exp2, the hardware's accuracy)
    against the reference-order form.  |variant 0 - variant v| IS an fp32 band of these formulas on this scene; the model the GPU
    tests use — decision margins for pixels, noise + flip budget for gradient rows — must bound it on EVERY pixel and EVERY row
    with every constant divided by K_BAND.  No escape clause: nothing is allowed to exceed the model here."""
    mk, kind, W, H, c2w, kw = BAND_SCENES[name]
    sc = mk()
    view = make_view(kind, W, H, c2w(), **kw)
    cam = view["oracle_cam"]
    d12, sph = scenes.pack_density(sc), sc["features"]
    rng = np.random.default_rng(11)
    rg = rng.normal(size=(H, W, 4)).astype(np.float32)
    dg = (0.1 * rng.normal(size=(H, W, 1))).astype(np.float32)
    A = oracle.forward(cam, W, H, d12, sph, view["ro"], view["rd"])
    margins2, pixel_budget = oracle.render_margins(cam, A, budget_bound=ROW_FLIP_BOUND)
    margins = margins2.min(-1)
    gA, sA, _, budget = oracle.backward(cam, A, rg, dg, flip_bound=ROW_FLIP_BOUND)
    report = {}
    for v in (1, 2):
        with oracle.variant(v):
            B = oracle.forward(cam, W, H, d12, sph, view["ro"], view["rd"])
            gB, sB, _ = oracle.backward(cam, B, rg, dg)
        # the binning is not part of the experiment: identical lists
        assert np.array_equal(A["sorted_ids"], B["sorted_ids"]) and np.array_equal(A["proj_pos"], B["proj_pos"])
        # pixels: whatever differs beyond the colour tolerance, or in its hit count, sits within FLIP_MARGIN_BOUND / K_BAND noise
        # widths of a threshold by the oracle's own margin estimate
        diff = np.abs(A["rgba"] - B["rgba"]).max(-1)
        out = (diff > COLOUR_TOL) | (A["hits"] != B["hits"])[..., 0]
        worst_margin = float(margins[out].max()) if out.any() else 0.0
        assert worst_margin <= FLIP_MARGIN_BOUND / K_BAND, (name, v, worst_margin)
        worst_calm_pixel = float(diff[margins >= FLIP_MARGIN_BOUND / K_BAND].max())
        assert worst_calm_pixel <= COLOUR_TOL / 2.0, (name, v, worst_calm_pixel)    # the flat colour tolerance: measured 8.7e-5 of 2e-4
        # ... and, quantitatively, every pixel stays within COLOUR_TOL + PIX_FLIP x its own flip budget (what its near-threshold entries
        # can move it by), hit counts within the number of such entries — with K_BAND to spare
        ddist = np.abs(A["dist"] - B["dist"])[..., 0] / max(1.0, float(np.abs(A["dist"]).max()))
        worst_budget = float((np.maximum(diff, ddist) / (COLOUR_TOL + PIX_FLIP * pixel_budget[..., 0])).max())
        assert worst_budget <= 0.5, (name, v, worst_budget)     # (a calm pixel at 8.7e-5 of the flat 2e-4 sets this: 0.43)
        assert bool((np.abs(A["hits"] - B["hits"])[..., 0] <= pixel_budget[..., 1]).all())
        rep = dict(pixel_outliers=int(out.sum()), worst_flip_margin=round(worst_margin, 2), worst_calm_pixel=worst_calm_pixel,
                   worst_pixel_vs_budget=round(worst_budget, 3))
        # gradient rows
        for j, (block, sl) in enumerate(GRAD_BLOCKS):
            a, b = gA[:, sl], gB[:, sl]
            nr, err = np.linalg.norm(a, axis=1), np.linalg.norm(a - b, axis=1)
            scale = float(np.quantile(nr[nr > 0], 0.99))
            noise, flip = budget[:, 5 + j], budget[:, j]
            calm = (flip == 0) & (nr > 0)
            # rows no flip-prone decision touches: the fp32 noise estimate alone (no relative or absolute floor) bounds the band
            r_noise = float((err[calm] / np.maximum(noise[calm], 1e-300)).max())
            assert r_noise <= ROW_NOISE / K_BAND, (name, v, block, r_noise)
            full = ROW_REL * nr + ROW_ABS * scale + ROW_NOISE * noise + ROW_FLIP * flip
            r_full = float((err / np.maximum(full, 1e-300)).max())
            assert r_full <= 1.0 / K_BAND, (name, v, block, r_full)
            rep[block] = (round(r_noise, 3), round(r_full, 3))
        # the SH rows: Y(dir) (x) dL/dRGB, budget columns 4 / 9 with the basis norm (tests/common.check_gradients_per_row)
        y = math.sqrt(16 / (4 * math.pi))
        nr, err = np.linalg.norm(sA, axis=1), np.linalg.norm(sA - sB, axis=1)
        scale = float(np.quantile(nr[nr > 0], 0.99))
        full = ROW_REL * nr + ROW_ABS * scale + ROW_NOISE * budget[:, 9] * y + ROW_FLIP * budget[:, 4] * y
        r_full = float((err / np.maximum(full, 1e-300)).max())
        assert r_full <= 1.0 / K_BAND, (name, v, "sh", r_full)
        rep["sh"] = round(r_full, 3)
        report[v] = rep
    print(f"[band {name}] {report}")


KERNEL_DEGREES = [0, 1, 3, 4, 5, 8]


def test_generalised_kernel_known_answers():
    """render.particle_kernel_degree: the response of every generalised Gaussian the reference compiles (gaussianParticles.cuh:256-306)
    at known arguments — s_n = -4.5 / 3^n makes every kernel's response exp(-4.5) at d2 = 9 (three "sigma"), the linear hat reaches
    zero at sqrt(d2) = 1 / 0.3296 — through the oracle's per-ray debug export on a one-Gaussian scene."""
    for n in KERNEL_DEGREES + [2]:
        prm = oracle.default_params()
        prm.kernel_degree = n
        sc = {"positions": np.zeros((1, 3), np.float32), "rotation": np.array([[1, 0, 0, 0]], np.float32),
              "scale": np.full((1, 3), 0.5, np.float32), "density": np.array([[0.9]], np.float32), "features": np.zeros((1, 48), np.float32)}
        W = H = 32
        view = make_view("pinhole", W, H, cams.look_at_c2w((0, 0, -4.0), (0, 0, 0)), fx=40.0)
        f = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"], params=prm)
        # the ray through pixel (px, 16) passes the centre at distance r = 4 * |dx| / fx (to first order): d2 = (r / 0.5)^2
        for px in (16, 18, 21, 25):
            dbg = oracle.debug_ray(view["oracle_cam"], f, px, 16, params=prm)
            if not len(dbg):
                continue
            d2 = float(dbg[0, 1])
            sn = prt._KERNEL_S[n]
            want = max(1.0 - sn * math.sqrt(d2), 0.0) if n == 0 else math.exp(-sn * d2 ** (0.5 * n))
            assert abs(float(dbg[0, 2]) - want) <= 2e-6 * max(want, 1e-3), (n, px, d2)
    for n in (1, 3, 4, 5, 8, 2):
        assert abs(prt._KERNEL_S[n] * 9.0 ** (0.5 * n) - 4.5) <= 1e-9 * 4.5 * 3 ** n


@pytest.mark.parametrize("degree", KERNEL_DEGREES)
def test_generalised_kernels_match_float64_autograd(degree):
    """The oracle's forward and backward for render.particle_kernel_degree != 2 against the float64 restatement: colours, hit counts,
    and every parameter gradient with the response's backward taken AS THE REFERENCE WRITES IT (particleResponseGrd<n>,
    gaussianParticles.cuh:211-254).  For every degree but 1 that is also the true derivative (plain autograd agrees); for degree 1 the
    reference multiplies by sqrt(d2) where the derivative divides by it — restated, and this test shows the two differ."""
    sc = scenes.scene_c1(250, seed=5)
    W, H = 48, 40
    view = make_view("pinhole", W, H, cams.look_at_c2w((0.3, 0.2, -3.5), (0, 0, 0)), fx=50)
    prm = oracle.default_params()
    prm.kernel_degree = degree
    f = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"], params=prm)
    f2 = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
    assert np.array_equal(f["sorted_keys"], f2["sorted_keys"])   # projection and binning do not know the kernel (gutProjector.cuh)
    assert np.abs(f["rgba"] - f2["rgba"]).max() > 1e-2           # ... the compositor does
    rg = np.random.default_rng(0).normal(size=(H, W, 4)).astype(np.float32)
    dg, sg, _ = oracle.backward(view["oracle_cam"], f, rg, np.zeros((H, W, 1), np.float32), params=prm)
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    grads = {}
    for as_written in (True, False):
        params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sc.items()}
        rgba, dist, hits = prt.render_tiled(params, view["tq"], W, H, view["ro"], view["rd"], f["tile_ranges"], f["sorted_ids"],
                                            kernel_degree=degree, reference_response_grad=as_written)
        assert np.abs(rgba.detach().numpy() - f["rgba"]).max() <= 2e-5
        assert np.abs(hits.numpy() - f["hits"][..., 0]).max() == 0
        (rgba * torch.tensor(rg, dtype=torch.float64)).sum().backward()
        grads[as_written] = params
    p = grads[True]
    tol = 5e-5 if degree != 0 else 5e-4   # the hat's 1 / sqrt(d2) amplifies fp32 rounding near the centre
    assert rel(p["positions"].grad.numpy(), dg[:, 0:3]) <= tol
    assert rel(p["density"].grad.numpy()[:, 0], dg[:, 3]) <= tol
    assert rel(p["rotation"].grad.numpy(), dg[:, 4:8]) <= tol
    assert rel(p["scale"].grad.numpy(), dg[:, 8:11]) <= tol
    assert rel(p["features"].grad.numpy(), sg) <= tol
    true_vs_written = rel(grads[False]["positions"].grad.numpy(), p["positions"].grad.numpy())
    if degree == 1:
        assert true_vs_written > 0.05, "degree 1: the reference's response gradient is NOT the derivative"
    else:
        assert true_vs_written <= 1e-9


def test_rolling_shutter_iterations_and_hit_count_options():
    """splat.n_rolling_shutter_iterations (cameraProjections.cuh:174: the loop count of projectPointWithShutter) and enable_hitcounts
    (rayPayload.cuh:126-128) in the oracle: fewer iterations move the projection of a moving camera, more than enough change nothing;
    without hit counts that output stays zero and nothing else changes."""
    sc = scenes.scene_c1(400, seed=3)
    W, H = 64, 48
    view = make_view("pinhole", W, H, cams.look_at_c2w((0.2, 0.1, -3.5), (0, 0, 0)), fx=60)
    pose_mod = importlib.import_module("3dgrut_amd.pose")
    tq_end = pose_mod.sensor_pose_from_c2w(cams.look_at_c2w((0.5, 0.0, -3.3), (0.1, 0, 0))).T_world_sensors[0]
    view["oracle_cam"] = dict(view["oracle_cam"], shutter=0, pose_end=tq_end)
    out = {}
    for it in (0, 1, 5, 9):
        prm = oracle.default_params()
        prm.rolling_shutter_iterations = it
        out[it] = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"], params=prm)
    assert not np.array_equal(out[0]["proj_pos"], out[5]["proj_pos"]) and not np.array_equal(out[1]["proj_pos"], out[5]["proj_pos"])
    cnt = (out[5]["tiles_count"] > 0) & (out[9]["tiles_count"] > 0)
    assert np.abs(out[9]["proj_pos"][cnt] - out[5]["proj_pos"][cnt]).max() <= 1e-3   # converged
    prm = oracle.default_params()
    prm.enable_hitcounts = 0
    off = oracle.forward(view["oracle_cam"], W, H, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"], params=prm)
    assert out[5]["hits"].max() > 0 and off["hits"].max() == 0
    assert np.array_equal(off["rgba"], out[5]["rgba"]) and np.array_equal(off["dist"], out[5]["dist"])
