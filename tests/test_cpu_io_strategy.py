"""CPU tests for the "next" rows N3 (densification logic, torch part) and N4 (PLY I/O, checkpoint keys)."""
import importlib
import os

import numpy as np
import pytest
import torch

from tests.common import scenes

io_ply = importlib.import_module("3dgrut_amd.io_ply")
native = importlib.import_module("3dgrut_amd.native")
strategy = importlib.import_module("3dgrut_amd.strategy")


def test_ply_round_trip_and_channel_major_layout(tmp_path):
    sc = scenes.scene_c1(123, 4)
    m = native.NativeGaussianModel(sc, device="cpu")
    path = os.path.join(tmp_path, "scene.ply")
    io_ply.export_native_model(m, path)
    d = io_ply.read_ply(path)
    raw = m.raw.numpy()
    assert np.array_equal(d["positions"], raw[:, 0:3]) and np.array_equal(d["density_logit"], raw[:, 3:4])
    assert np.array_equal(d["rotation_raw"], raw[:, 4:8]) and np.array_equal(d["log_scale"], raw[:, 8:11])
    assert np.array_equal(d["features48"], m.features.numpy())
    # file layout is 3DGS-compatible: f_rest_k holds channel (k // 15), coefficient 1 + k % 15
    with open(path, "rb") as fh:
        blob = fh.read()
    header_end = blob.index(b"end_header\n") + len(b"end_header\n")
    table = np.frombuffer(blob[header_end:], "<f4").reshape(123, 62)
    f = m.features.numpy().reshape(123, 16, 3)
    assert np.array_equal(table[:, 6:9], f[:, 0, :])
    for k in (0, 14, 15, 44):
        assert np.array_equal(table[:, 9 + k], f[:, 1 + k % 15, k // 15])
    assert np.array_equal(table[:, 54], raw[:, 3])
    sc2 = io_ply.scene_from_ply(path)
    assert np.allclose(sc2["scale"], sc["scale"], rtol=1e-6) and np.allclose(sc2["density"], sc["density"], atol=1e-6)


def test_checkpoint_dict_keys():
    m = native.NativeGaussianModel(scenes.scene_c1(10, 1), device="cpu")
    ck = io_ply.checkpoint_dict(m)
    for key in ("positions", "rotation", "scale", "density", "features_albedo", "features_specular", "n_active_features", "max_n_features"):
        assert key in ck
    assert ck["features_specular"].shape == (10, 45) and ck["density"].shape == (10, 1)


class _FakeStepper:
    """NativeTrainStep state without the GPU pieces (strategy code only touches tensors)."""
    def __init__(self, model):
        self.model = model
        n = model.num_gaussians
        self.m12, self.v12 = torch.ones(n, 12), torch.ones(n, 12)
        self.m48, self.v48 = torch.ones(n, 48), torch.ones(n, 48)
        self.resized = 0
    def resize_workspace(self):
        self.resized += 1


def test_gs_clone_split_prune_reset():
    sc = scenes.scene_c1(200, 8)
    sc["scale"][:100] = 0.001   # small -> clone candidates
    sc["scale"][100:] = 0.5     # large -> split candidates
    m = native.NativeGaussianModel(sc, device="cpu")
    st = _FakeStepper(m)
    gs = strategy.GSStrategy(st, seed=3)
    grad = torch.zeros(200, 3); grad[:50] = 1.0; grad[100:130] = 1.0   # only these exceed the threshold
    gs.update_gradient_buffer(grad, torch.tensor([0.0, 0.0, -4.0]))
    assert int(gs.grad_norm_denom.sum()) == 80
    raw0 = m.raw.clone()
    gs.densify(scene_extent=1.0, step=5)
    # 50 clones appended, then 30 large ones replaced by 2 children each: 200 + 50 - 30 + 60
    assert m.num_gaussians == 280 and st.m12.shape == (280, 12) and st.v48.shape == (280, 48)
    assert float(st.m12[-60:].abs().max()) == 0 and float(st.m48[-60:].abs().max()) == 0      # new rows start with zero moments
    assert torch.allclose(torch.exp(m.raw[-60:, 8:11]), torch.exp(raw0[100:130, 8:11]).repeat(2, 1) / 1.6, rtol=1e-5)
    assert int(gs.grad_norm_denom.sum()) == 0 and gs.grad_norm_accum.shape == (280, 1)
    # determinism across "ranks": same seed/step -> identical children
    m2 = native.NativeGaussianModel(sc, device="cpu"); st2 = _FakeStepper(m2); gs2 = strategy.GSStrategy(st2, seed=3)
    gs2.update_gradient_buffer(grad, torch.tensor([0.0, 0.0, -4.0])); gs2.densify(1.0, step=5)
    assert torch.equal(m.raw, m2.raw)
    # prune + reset
    m.raw[:7, 3] = -10.0
    assert gs.prune_opacity() == 7 and m.num_gaussians == 273
    gs.reset_density()
    assert float(torch.sigmoid(m.raw[:, 3]).max()) <= 0.01 + 1e-6 and float(st.m12[:, 3].abs().max()) == 0


# ---------------------------------------------------------------------------------------------------
# COLMAP sparse models (N4)
# ---------------------------------------------------------------------------------------------------
def _synthetic_colmap(root, n_images=9, n_points=500):
    import importlib
    io_colmap = importlib.import_module("3dgrut_amd.io_colmap")
    cams_mod = importlib.import_module("3dgrut_amd.cameras")
    rng = np.random.default_rng(0)
    cameras = {1: io_colmap.ColmapCamera(1, "PINHOLE", 640, 480, np.array([500.0, 510.0, 320.0, 240.0])),
               2: io_colmap.ColmapCamera(2, "SIMPLE_PINHOLE", 320, 240, np.array([260.0, 160.0, 120.0])),
               3: io_colmap.ColmapCamera(3, "OPENCV_FISHEYE", 400, 300, np.array([180.0, 181.0, 200.0, 150.0, 0.02, -0.004, 0.001, 0.0]))}
    images, c2ws = [], []
    for i in range(n_images):
        c2w = cams_mod.orbit_c2w(4.0, 40.0 * i, 10.0).astype(np.float64)
        w2c = np.linalg.inv(c2w)
        images.append(io_colmap.ColmapImage(i + 1, io_colmap.rotation_to_qvec(w2c[:3, :3]), w2c[:3, 3], 1 + i % 3, f"img_{i:03d}.png"))
        c2ws.append(c2w)
    xyz = rng.normal(size=(n_points, 3))
    rgb = rng.integers(0, 256, size=(n_points, 3)).astype(np.uint8)
    io_colmap.write_model_binary(os.path.join(root, "sparse", "0"), cameras, images, xyz, rgb)
    return io_colmap, cameras, images, np.stack(c2ws), xyz, rgb


def test_colmap_binary_round_trip_and_scene(tmp_path):
    io_colmap, cameras, images, c2ws, xyz, rgb = _synthetic_colmap(str(tmp_path))
    sparse = os.path.join(str(tmp_path), "sparse", "0")
    cams_r = io_colmap.read_cameras_binary(os.path.join(sparse, "cameras.bin"))
    assert set(cams_r) == {1, 2, 3} and cams_r[3].model == "OPENCV_FISHEYE" and np.array_equal(cams_r[3].params, cameras[3].params)
    ims_r = io_colmap.read_images_binary(os.path.join(sparse, "images.bin"))
    assert [im.name for im in ims_r] == [im.name for im in images]
    assert np.allclose(ims_r[4].qvec, images[4].qvec) and np.allclose(ims_r[4].tvec, images[4].tvec)
    p, c, _ = io_colmap.read_points3D_binary(os.path.join(sparse, "points3D.bin"))
    assert np.allclose(p, xyz) and np.array_equal(c, rgb)
    train = io_colmap.ColmapScene(str(tmp_path), split="train", test_split_interval=8)
    test = io_colmap.ColmapScene(str(tmp_path), split="test", test_split_interval=8)
    assert len(train) == 7 and len(test) == 2 and [im.name for im in test.images] == ["img_000.png", "img_008.png"]
    assert np.abs(train.poses[0].astype(np.float64) - c2ws[1]).max() <= 1e-5        # C2W = inv([R|t])
    centre = c2ws[[1, 2, 3, 4, 5, 6, 7], :3, 3].mean(0)
    assert abs(train.cameras_extent - 1.1 * np.linalg.norm(c2ws[[1, 2, 3, 4, 5, 6, 7], :3, 3] - centre, axis=1).max()) <= 1e-4
    # per-view batch: rays and intrinsics in the boundary's formats, pose on the host
    b = train.batch(0, device="cpu")                                                  # image 2 -> camera 2 (SIMPLE_PINHOLE)
    assert b.rays_dir.shape == (1, 240, 320, 3) and b.intrinsics_OpenCVPinholeCameraModelParameters["focal_length"][0] == 260.0
    assert not b.T_to_world.is_cuda and b.rgb_gt is None
    fb = train.batch(1, device="cpu")                                                 # image 3 -> camera 3 (fisheye)
    K = fb.intrinsics_OpenCVFisheyeCameraModelParameters
    assert fb.rays_dir.shape == (1, 300, 400, 3) and abs(float(K["radial_coeffs"][0]) - 0.02) < 1e-7 and K["max_angle"] > 1.0
    assert np.abs(np.linalg.norm(fb.rays_dir.numpy(), axis=-1) - 1.0).max() <= 1e-5
    half = io_colmap.ColmapScene(str(tmp_path), downsample_factor=2)
    hb = half.batch(2, device="cpu")                                                  # image 4 -> camera 1 (PINHOLE) at half size
    assert hb.rays_dir.shape == (1, 240, 320, 3) and hb.intrinsics_OpenCVPinholeCameraModelParameters["focal_length"][1] == 255.0
    # initial Gaussians from the SfM points
    g = train.initial_gaussians()
    assert g["positions"].shape == (500, 3) and g["features"].shape == (500, 48) and np.all(g["density"] == 0.1)
    d = np.linalg.norm(xyz[:, None, :] - train.camera_centers[None], axis=-1).min(1)
    assert np.allclose(g["scale"][:, 0], 0.01 * d, rtol=1e-5)
    assert np.allclose(g["features"][:, :3] * 0.28209479177387814 + 0.5, rgb / 255.0, atol=1e-6)
    gk = train.initial_gaussians(use_observation_points=False)
    assert np.all(gk["scale"] > 0) and not np.allclose(gk["scale"], g["scale"])


def test_colmap_text_readers_and_unsupported_model(tmp_path):
    import importlib
    io_colmap = importlib.import_module("3dgrut_amd.io_colmap")
    sparse = os.path.join(str(tmp_path), "sparse", "0")
    os.makedirs(sparse)
    open(os.path.join(sparse, "cameras.txt"), "w").write("# Camera list\n1 PINHOLE 100 80 90.0 91.0 50.0 40.0\n")
    open(os.path.join(sparse, "images.txt"), "w").write("# Image list\n1 1 0 0 0 0.1 0.2 3.0 1 a b.png\n1.0 2.0 -1\n2 1 0 0 0 0 0 3.5 1 c.png\n\n")
    open(os.path.join(sparse, "points3D.txt"), "w").write("# pts\n7 0.5 0.25 1.0 10 20 30 0.1 1 2\n")
    sc = io_colmap.ColmapScene(str(tmp_path), test_split_interval=0)
    assert len(sc) == 2 and sc.images[0].name == "a b.png" and sc.images[1].tvec[2] == 3.5
    assert np.allclose(sc.poses[0][:3, 3], [-0.1, -0.2, -3.0])
    xyz, rgb = sc.points()
    assert np.allclose(xyz, [[0.5, 0.25, 1.0]]) and rgb.tolist() == [[10, 20, 30]]
    open(os.path.join(sparse, "cameras.txt"), "w").write("1 OPENCV 100 80 90 91 50 40 0.1 0 0 0\n")
    with pytest.raises(ValueError):
        io_colmap.ColmapScene(str(tmp_path))


def test_fisheye_rays_invert_the_forward_polynomial():
    import importlib
    cams_mod = importlib.import_module("3dgrut_amd.cameras")
    k = (0.03, -0.006, 0.0012, -0.0001)
    W, H, f = 200, 150, 90.0
    _, rd = cams_mod.fisheye_rays(W, H, f, f, radial=k)
    d = rd[0].astype(np.float64)
    theta = np.arccos(np.clip(d[..., 2], -1, 1))
    t2 = theta * theta
    delta = theta * (1 + t2 * (k[0] + t2 * (k[1] + t2 * (k[2] + t2 * k[3]))))
    x, y = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5, indexing="xy")
    want = np.sqrt(((x - W / 2) / f) ** 2 + ((y - H / 2) / f) ** 2)
    assert np.abs(delta - want).max() <= 2e-4      # three Newton steps (the reference's setting) from the linear guess


def test_morton_order_is_a_locality_preserving_permutation():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, size=(4096, 3))
    perm = scenes.morton_order(pts)
    assert sorted(perm.tolist()) == list(range(4096))
    d_sorted = np.linalg.norm(np.diff(pts[perm], axis=0), axis=1).mean()
    d_random = np.linalg.norm(np.diff(pts, axis=0), axis=1).mean()
    assert d_sorted < 0.25 * d_random
    sc = scenes.reorder(scenes.scene_c1(100, 0), scenes.morton_order(scenes.scene_c1(100, 0)["positions"]))
    assert sc["features"].shape == (100, 48)


def test_spatial_permutation_matches_the_numpy_morton_order():
    import torch
    native = importlib.import_module("3dgrut_amd.native")
    pts = np.random.default_rng(3).uniform(-5, 7, size=(5000, 3)).astype(np.float32)
    perm = native.spatial_permutation(torch.as_tensor(pts)).numpy()
    assert np.array_equal(perm, scenes.morton_order(pts))
    assert np.array_equal(np.sort(perm), np.arange(5000))


def test_bench_takes_a_real_scene_as_ply_and_colmap(tmp_path):
    """bench.py --ply PATH [--colmap DIR]: the scene the bench builds its model from is the PLY's, row for row, the views are the
    COLMAP directory's training split, and the line says so in `data` (real scenes substitute the stand-ins 1:1)."""
    import argparse
    import bench
    sc = scenes.scene_c1(300, 5)
    m = native.NativeGaussianModel(sc, device="cpu")
    ply = os.path.join(str(tmp_path), "point_cloud.ply")
    io_ply.export_native_model(m, ply)
    _synthetic_colmap(str(tmp_path))
    args = argparse.Namespace(ply=ply, colmap=str(tmp_path), colmap_downsample=1, workload="bicycle_like_6M_1237x822")
    scene, colmap, extent = bench.load_scene(args, "bicycle_like_6M_1237x822")
    assert scene["positions"].shape == (300, 3) and np.allclose(scene["positions"], sc["positions"])
    assert np.allclose(scene["scale"], sc["scale"], rtol=1e-5) and np.allclose(scene["density"], np.clip(sc["density"], 1e-6, 1 - 1e-6), rtol=1e-4, atol=1e-6)
    assert np.allclose(scene["features"], sc["features"])
    assert len(colmap) == 7 and abs(extent - colmap.cameras_extent) == 0 and extent > 0
    b = colmap.batch(0, device="cpu")
    assert b.rays_dir.shape[0] == 1 and b.rays_dir.shape[3] == 3
    assert bench.data_label(args) == "ply:point_cloud.ply colmap:" + os.path.basename(os.path.normpath(str(tmp_path)))
    # without --ply: the seeded stand-in of the named workload (a tiny one here)
    plain = argparse.Namespace(ply=None, colmap=None, workload="c1_1k_128x128")
    scene2, colmap2, extent2 = bench.load_scene(plain, "c1_1k_128x128")
    assert scene2["positions"].shape == (1000, 3) and colmap2 is None and extent2 == 1.0 and bench.data_label(plain) == "synthetic"
    # the sensitivity workload of a --ply run is not the PLY
    assert bench.load_scene(argparse.Namespace(ply=ply, colmap=None, workload="c1_1k_128x128"), "lego_like_300k_800x800", dict(n=10, seed=1))[0]["positions"].shape[0] == 10
