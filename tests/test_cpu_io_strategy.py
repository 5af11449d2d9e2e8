"""CPU tests for the "next" rows N3 (densification logic, torch part) and N4 (PLY I/O, checkpoint keys)."""
import importlib
import os

import numpy as np
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
