"""Randomised parity sweep: seeded random scenes / cameras / resolutions / SH degrees against the CPU oracle.
Same bars as tests/test_gpu_parity.py: integer buffers and projection floats bit-exact, image within 2e-4 (threshold flips
allowed on <= 0.1 % of the pixels, never above 1e-2), gradients within 3e-3 relative L2 per parameter block — except the
geometry blocks (position / rotation / scale) of the deliberately ill-conditioned draws (scales down to 2e-4 scene units,
20:1 needles): there the ray origin in the particle's canonical space is ~1e4 while the response depends on differences of
order 1, fp32 loses three to four digits in ANY formulation (the fp32 oracle itself sits 2e-4 from a float64 autograd there,
the kernel 3e-3 .. 7e-3), and the bar is 1.5e-2."""
import importlib

import numpy as np
import pytest
import torch

from tests.common import FISHEYE_DIST, ROW_FLIP_BOUND, cams, check_colour_outliers, make_view, rel_l2, scenes
from tests.test_gpu_parity import DIST, _activated_grads, _oracle_inputs, _run_gpu

pytestmark = pytest.mark.gpu
oracle = importlib.import_module("oracle.oracle")


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 3000))
    sc = scenes.scene_c1(n, seed=int(rng.integers(0, 1 << 30)))
    sc["scale"] = (sc["scale"] * float(rng.choice([0.2, 0.5, 1.0, 2.0, 5.0]))).astype(np.float32)
    if rng.random() < 0.3:   # needle-like Gaussians
        sc["scale"][:, int(rng.integers(0, 3))] *= 0.05
    W, H = int(rng.integers(17, 200)), int(rng.integers(17, 160))
    kind = str(rng.choice(["pinhole", "pinhole_dist", "fisheye"]))
    r = float(rng.uniform(0.3, 5.0))
    eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * r
    tgt = rng.uniform(-0.3, 0.3, size=3)
    kw = {}
    if kind.startswith("pinhole"):
        kw["fx"] = float(rng.uniform(0.4, 1.5) * W)
        kw["fy"] = kw["fx"] * float(rng.uniform(0.9, 1.1))
        if kind == "pinhole_dist":
            kw["distortion"] = DIST
    sh = int(rng.integers(0, 4))
    if kind == "fisheye":
        # drawn AFTER everything else so that the twelve round-3 configurations keep their scenes and cameras: two in three fisheye
        # cameras carry a non-zero polynomial (k1..k4 up to a ScanNet++ DSLR's size), one in three also a cone inside the image
        u = float(rng.random())
        if u < 2 / 3:
            kw["distortion"] = dict(radial=[float(k * rng.uniform(0.3, 1.5)) for k in FISHEYE_DIST["radial"]])
            if u < 1 / 3:
                kw["distortion"]["max_angle"] = float(rng.uniform(0.4, 0.9))
    view = make_view("fisheye" if kind == "fisheye" else "pinhole", W, H, cams.look_at_c2w(tuple(eye), tuple(tgt)), **kw)
    return sc, view, W, H, sh, rng


def _variant(seed):
    """seeds >= 100: the same draw of scene and camera (seed - 100) under a drawn render-config variant — one of the reference's other
    generalised Gaussian kernels (particle_kernel_degree), sometimes other thresholds (incl. a max alpha the unsorted backward ignores,
    gaussianParticles.cuh:528) and hit counts off."""
    if seed < 100:
        return {}, {}
    rng = np.random.default_rng(7000 + seed)
    degree = int(rng.choice([0, 1, 3, 4, 5, 8]))
    conf, prm = {"particle_kernel_degree": degree}, {"kernel_degree": degree}
    if rng.random() < 0.5:
        mr, ma, mx = float(rng.choice([0.005, 0.03])), float(rng.choice([0.002, 0.01])), float(rng.choice([0.9, 0.995]))
        conf.update(particle_kernel_min_response=mr, particle_kernel_min_alpha=ma, particle_kernel_max_alpha=mx)
        prm.update(min_kernel_density=mr, alpha_threshold=ma, max_alpha=mx)
    if rng.random() < 0.25:
        conf["enable_hitcounts"] = False
        prm["enable_hitcounts"] = 0
    return conf, prm


@pytest.mark.parametrize("seed", list(range(16)) + list(range(100, 112)))
def test_random_configuration_matches_the_oracle(seed):
    sc, view, W, H, sh, rng = _case(seed % 100)
    conf, overrides = _variant(seed)
    prm = oracle.default_params()
    for k, v in overrides.items():
        assert hasattr(prm, k), k
        setattr(prm, k, v)
    model, d12, sph = _oracle_inputs(sc, sh)
    ref = oracle.forward(view["oracle_cam"], W, H, d12, sph, view["ro"], view["rd"], sh_degree=sh, params=prm)
    rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
    dist_grad = (0.1 * rng.normal(size=(H, W, 1))).astype(np.float32) if seed % 3 == 0 else None
    res = _run_gpu(sc, view, sh, rgba_grad=rgba_grad, dist_grad=dist_grad, model=model, render_conf=conf)
    raster = res["tracer"].tracer_wrapper
    for key in ("tiles_count", "tiles_offset", "unsorted_ids", "sorted_ids"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint32), ref[key]), key
    for key in ("unsorted_keys", "sorted_keys"):
        assert np.array_equal(raster.debug_buffer(key).cpu().numpy().view(np.uint64), ref[key]), key
    for key in ("proj_pos", "conic_opacity", "extent", "depth", "feat"):
        got = raster.debug_buffer(key).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.ascontiguousarray(ref[key]).reshape(-1).view(np.uint32)), key
    out = res["out"]
    rgba = np.concatenate([out["pred_rgb"][0].detach().cpu().numpy(), out["pred_opacity"][0].detach().cpu().numpy()], -1)
    if ref["M"]:
        margins, pixel_budget = oracle.render_margins(view["oracle_cam"], ref, params=prm, budget_bound=ROW_FLIP_BOUND)
        # quantitative form: the ill-conditioned draws (needles, 2e-4 scales) have wide noise bands — a large share of their pixels has
        # some decision near a threshold, and each is allowed what those decisions can move it by
        check_colour_outliers(rgba, out["hits_count"][0].detach().cpu().numpy(), ref, margins, label=f"fuzz {seed}", budget=pixel_budget)
    else:
        assert np.abs(rgba - ref["rgba"]).max() == 0.0
    st = raster.stats()
    assert st["traversed_fwd"] == ref["traversed_fwd"]
    if ref["M"] == 0:
        return
    dens_g, sph_g, _ = oracle.backward(view["oracle_cam"], ref, rgba_grad,
                                       dist_grad if dist_grad is not None else np.zeros((H, W, 1), np.float32), params=prm)
    exp = _activated_grads(res["model"], dens_g, sph_g)
    for k, e in exp.items():
        g = getattr(res["model"], k).grad.cpu().numpy()
        if np.linalg.norm(e) < 1e-12:
            assert np.abs(g).max() <= 1e-9, k
            continue
        tol = 1.5e-2 if k in ("positions", "rotation", "scale") else 3e-3
        assert rel_l2(g, e) <= tol, f"seed {seed} {k}: rel L2 {rel_l2(g, e)}"
    assert raster.stats()["traversed_bwd"] == ref["traversed_bwd"]
