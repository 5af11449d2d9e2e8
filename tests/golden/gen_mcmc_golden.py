"""Generate a golden fixture for the MCMC strategy ("next" row N3) from the reference's own Python.

Runs ONLY in the build container (needs /root/reference); nothing of the reference travels — the output is data
(inputs + what the reference's code left behind for them):

    tests/golden/mcmc_golden.npz

What is driven: threedgrut/strategy/mcmc.py `MCMCStrategy` (:48-197: relocate_gaussians, add_new_gaussians, perturb_gaussians,
sample_new_gaussians) on top of threedgrut/strategy/base.py `BaseStrategy._update_param_with_optimizer` (:52-83) and
threedgrut/utils/misc.py (`_multinomial_sample`, `quaternion_to_so3`, `check_step_condition`, the inverse activations), imported by
file path, with a small fake `MixtureOfGaussians` (raw nn.Parameters named as the reference names them, the reference's
activations, `get_covariance` as model.py:95-105 builds it from the reference's `quaternion_to_so3`, and a real torch.optim.Adam
with one named group per parameter, whose state the strategy edits).  After each stage every parameter and both Adam moments are
recorded, together with everything random or external that went in (below).

Environment adaptations (none changes what mcmc.py computes with what it is given):
  * packages mcmc.py imports but does not use for the code under test are module shells (threedgrut.model.model: a type
    annotation; threedgrut.utils.logger: statistics, switched off by `print_stats: false`; omegaconf / tensorboard, pulled in by
    utils/misc.py);
  * `load_mcmc_plugin()` would JIT-build the CUDA extension lib_mcmc_cc (strategy/src/gaussian_mcmc.cu), which cannot exist here:
    it is a no-op, and `_mcmc_plugin.compute_relocation_tensor` is a RECORDER — it stores the three tensors mcmc.py hands to the
    kernel and returns the kernel's closed form (Eq. 9 of the MCMC paper as gaussian_mcmc.cu:33-73 writes it) evaluated in
    float64 and rounded to float32.  The fixture therefore pins the HOST logic of mcmc.py (what goes into the kernel, what is
    done with what comes out); the kernel itself is pinned by its closed form in tests/test_gpu_native.py, and the recorded
    outputs are compared with the HIP kernel's there too;
  * `_multinomial_sample` (torch.multinomial on the CPU generator here, on the CUDA generator in a real run) and
    `torch.randn_like` are wrapped so that their results are recorded: the draws are inputs of the fixture.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_host_golden import _load, _shell, load_reference  # noqa: E402
from gen_strategy_golden import NS  # noqa: E402

CONF = NS(dict(
    model=dict(density_activation="sigmoid"),
    strategy=dict(
        print_stats=False, binom_n_max=51, opacity_threshold=0.005,
        relocate=dict(start_iteration=500, end_iteration=25000, frequency=100),
        perturb=dict(start_iteration=0, end_iteration=27500, frequency=1, noise_lr=500000.0),
        add=dict(start_iteration=500, end_iteration=25000, frequency=100, max_n_gaussians=330))))

PARAMS = ("positions", "density", "features_albedo", "features_specular", "rotation", "scale")
POSITION_LR = 1.6e-4 * 0.37     # some point on the position schedule


def relocation_closed_form(opacities, scales, ratios, binoms, n_max):
    """gaussian_mcmc.cu:33-73 in float64: new_opacity = 1 - (1 - o)^(1/ratio); new_scale = o / denom * scale with
    denom = sum_{i=1..ratio} sum_{k=0..i-1} binom(i-1, k) (-1)^k new_opacity^(k+1) / sqrt(k+1)."""
    o = opacities.double().flatten()
    r = ratios.long().flatten()
    new_o = 1.0 - torch.pow(1.0 - o, 1.0 / r.double())
    denom = torch.zeros_like(o)
    b = binoms.double()
    for i in range(1, int(r.max()) + 1):
        live = (r >= i).double()
        for k in range(i):
            denom += live * b[i - 1, k] * ((-1.0) ** k) * torch.pow(new_o, k + 1) / math.sqrt(k + 1)
    coeff = o / denom
    return new_o.float().view_as(opacities), (coeff[:, None] * scales.double()).float()


class FakeMoG:
    """The slice of threedgrut/model/model.py:45-205 that mcmc.py touches."""
    def __init__(self, init, misc):
        self.device = "cpu"
        self._misc = misc
        for k in PARAMS:
            setattr(self, k, torch.nn.Parameter(torch.as_tensor(init[k]).clone()))
        groups = [dict(params=[getattr(self, k)], lr=(POSITION_LR if k == "positions" else 1e-3), name=k) for k in PARAMS]
        self.optimizer = torch.optim.Adam(groups, eps=1e-15)
        self.scale_activation, self.scale_activation_inv = misc.get_activation_function("exp"), misc.get_activation_function("exp", inverse=True)
        self.density_activation = misc.get_activation_function("sigmoid")
        self.density_activation_inv = misc.get_activation_function("sigmoid", inverse=True)
        self.rotation_activation = misc.get_activation_function("normalize")

    @property
    def num_gaussians(self):
        return self.positions.shape[0]

    def get_positions(self):
        return self.positions

    def get_scale(self):
        return self.scale_activation(self.scale)

    def get_density(self):
        return self.density_activation(self.density)

    def get_rotation(self):
        return self.rotation_activation(self.rotation)

    def get_covariance(self):                     # model.py:95-105
        scales = self.get_scale()
        S = torch.zeros((self.num_gaussians, 3, 3), dtype=scales.dtype, device=self.device)
        R = self._misc.quaternion_to_so3(self.get_rotation())
        S[:, 0, 0] = scales[:, 0]
        S[:, 1, 1] = scales[:, 1]
        S[:, 2, 2] = scales[:, 2]
        return R @ S @ S.transpose(1, 2) @ R.transpose(1, 2)


def snapshot(tag, model, out):
    for k in PARAMS:
        p = getattr(model, k)
        out[f"{tag}/{k}"] = p.detach().numpy().copy()
        st = model.optimizer.state[p]
        out[f"{tag}/{k}/exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"{tag}/{k}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()


def main():
    load_reference()
    misc = sys.modules["threedgrut.utils.misc"]
    _shell("threedgrut.model")
    _shell("threedgrut.model.model", MixtureOfGaussians=object)
    _shell("threedgrut.utils.logger", logger=types.SimpleNamespace(info=lambda *a, **k: None))
    _shell("threedgrut.strategy")
    _load("threedgrut.strategy.base", "threedgrut/strategy/base.py")
    mcmc = _load("threedgrut.strategy.mcmc", "threedgrut/strategy/mcmc.py")

    out = {}
    calls = dict(kernel=0, sample=0)

    class Plugin:
        @staticmethod
        def compute_relocation_tensor(opacities, scales, ratios, binoms, n_max):
            i = calls["kernel"]; calls["kernel"] += 1
            new_o, new_s = relocation_closed_form(opacities, scales, ratios, binoms, n_max)
            out[f"kernel{i}/opacities"], out[f"kernel{i}/scales"] = opacities.numpy().copy(), scales.numpy().copy()
            out[f"kernel{i}/ratios"] = ratios.numpy().copy()
            out[f"kernel{i}/new_opacities"], out[f"kernel{i}/new_scales"] = new_o.numpy().copy(), new_s.numpy().copy()
            assert ratios.dtype == torch.int32 and int(n_max) == CONF.strategy.binom_n_max
            return new_o, new_s

    mcmc.load_mcmc_plugin = lambda: None
    mcmc._mcmc_plugin = Plugin
    ref_sample = mcmc._multinomial_sample

    def recording_sample(probabilities, n, replacement=True):
        i = calls["sample"]; calls["sample"] += 1
        idx = ref_sample(probabilities, n, replacement=replacement)
        out[f"sample{i}/probabilities"], out[f"sample{i}/indices"] = probabilities.detach().numpy().copy(), idx.numpy().copy()
        return idx
    mcmc._multinomial_sample = recording_sample

    class TorchRecordingRandn:
        def __getattr__(self, k):
            return getattr(torch, k)

        def randn_like(self, t):
            u = torch.randn_like(t)
            out["perturb/unit_draws"] = u.numpy().copy()
            return u
    mcmc.torch = TorchRecordingRandn()

    rng = np.random.default_rng(20261)
    torch.manual_seed(20261)
    n = 300
    init = dict(
        positions=rng.uniform(-1, 1, size=(n, 3)).astype(np.float32),
        rotation=rng.normal(size=(n, 4)).astype(np.float32),
        scale=np.log(rng.uniform(0.004, 0.05, size=(n, 3))).astype(np.float32),
        density=rng.normal(-1.0, 3.0, size=(n, 1)).astype(np.float32),             # logits; sigmoid <= 0.005 below -5.29: some are dead
        features_albedo=rng.uniform(-1, 1, size=(n, 3)).astype(np.float32),
        features_specular=rng.normal(0, 0.1, size=(n, 45)).astype(np.float32))
    init["density"][:6, 0] = 6.0                                                   # a few near-opaque ones: drawn several times (ratios > 2)
    model = FakeMoG(init, misc)
    for k in PARAMS:   # Adam moments as after some training (non-zero everywhere)
        getattr(model, k).grad = torch.as_tensor(rng.normal(size=getattr(model, k).shape).astype(np.float32) * 1e-3)
    model.optimizer.step()
    model.optimizer.zero_grad()

    strat = mcmc.MCMCStrategy(CONF, model)
    out["binoms"] = strat.binoms.numpy().copy()
    out["position_lr"] = np.float64(POSITION_LR)
    snapshot("start", model, out)
    n_dead = int((model.get_density() <= CONF.strategy.opacity_threshold).sum())
    assert n_dead >= 10, n_dead
    out["n_dead"] = np.int64(n_dead)

    strat.relocate_gaussians()
    snapshot("after_relocate", model, out)
    strat.add_new_gaussians()
    snapshot("after_add", model, out)
    assert model.num_gaussians == min(CONF.strategy.add.max_n_gaussians, int(1.05 * n)) > n
    strat.perturb_gaussians()
    snapshot("after_perturb", model, out)
    strat.add_new_gaussians()                   # int(1.05 * 315) = 330 = the cap
    strat.add_new_gaussians()                   # at the cap: nothing to add, no kernel call
    snapshot("after_cap", model, out)
    assert model.num_gaussians == CONF.strategy.add.max_n_gaussians
    out["kernel_calls"], out["sample_calls"] = np.int64(calls["kernel"]), np.int64(calls["sample"])
    assert max(int(out[f"kernel{i}/ratios"].max()) for i in range(calls["kernel"])) >= 3

    # the schedule the trainer applies these with (utils/misc.py:198-202 on configs/strategy/mcmc.yaml)
    s = CONF.strategy
    steps = range(0, 28001)
    for name, c in (("relocate", s.relocate), ("add", s.add), ("perturb", s.perturb)):
        out[f"schedule_{name}"] = np.array([t for t in steps if misc.check_step_condition(t, c.start_iteration, c.end_iteration, c.frequency)], np.int32)

    np.savez_compressed(os.path.join(HERE, "mcmc_golden.npz"), **out)
    print("wrote mcmc_golden.npz; dead", n_dead, "Gaussians per stage:",
          {k.split("/")[0]: v.shape[0] for k, v in out.items() if k.endswith("/positions")}, "kernel calls", calls["kernel"])


if __name__ == "__main__":
    main()
