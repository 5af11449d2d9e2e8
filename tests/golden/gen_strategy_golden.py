"""Generate a golden fixture for the densification strategy ("next" row N3) from the reference's own Python.

Runs ONLY in the build container (needs /root/reference); nothing of the reference travels — the output is data
(inputs + what the reference's code left behind for them):

    tests/golden/strategy_golden.npz

What is driven: threedgrut/strategy/gs.py `GSStrategy` (:26-306) on top of threedgrut/strategy/base.py
`BaseStrategy._update_param_with_optimizer` (:52-83), imported by file path, with a small fake `MixtureOfGaussians`
(raw nn.Parameters named as the reference names them, the reference's activations exp / sigmoid / normalize, and a real
torch.optim.Adam with one named group per parameter, whose state the strategy edits).  The sequence is the one the
reference's trainer runs (trainer.py:741-760): three views' `update_gradient_buffer`, `densify_gaussians` (clone then
split), `prune_gaussians_opacity`, `decay_density`, `reset_density`; after each stage every parameter, both Adam moments and
the two densification buffers are recorded.

Environment adaptations (none changes what gs.py computes):
  * packages gs.py imports but does not use for the code under test are module shells: threedgrut.model.model
    (`MixtureOfGaussians` is a type annotation only), threedgrut.utils.logger (`logger.info` statistics, switched off by
    `print_stats: false`), omegaconf / tensorboard (pulled in by utils/misc.py);
  * `torch.cuda.nvtx.range` (decorators on the methods) is a no-op context manager: there is no GPU in this container;
  * gs.py:split_gaussians allocates two temporaries with a hard-coded device="cuda" (:133, :144): the module sees a `torch`
    whose zeros() maps that device string to "cpu";
  * `torch.normal(mean, std)` (:145) is replaced by `unit * std` with `unit` a recorded standard-normal tensor, so that the
    fixture pins WHERE the children go for known draws (the reference's own draws depend on the CUDA generator state).
"""
import importlib.util
import os
import sys
import types
from contextlib import contextmanager

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_host_golden import _load, _shell  # noqa: E402  (module shells + load-by-path helpers)


class NS(dict):
    """Attribute-access nested config (what gs.py does with its OmegaConf node)."""
    def __getattr__(self, k):
        v = self[k]
        return NS(v) if isinstance(v, dict) else v


CONF = NS(dict(
    model=dict(density_activation="sigmoid"),
    strategy=dict(
        print_stats=False,
        densify=dict(frequency=300, start_iteration=500, end_iteration=15000, clone_grad_threshold=0.0002, split_grad_threshold=0.0002,
                     relative_size_threshold=0.01, split=dict(n_gaussians=2)),
        prune=dict(frequency=100, start_iteration=500, end_iteration=15000, density_threshold=0.005),
        reset_density=dict(frequency=3000, start_iteration=0, end_iteration=15000, new_max_density=0.01),
        density_decay=dict(gamma=0.99, start_iteration=-1, end_iteration=-1, frequency=50),
        prune_weight=dict(frequency=100, start_iteration=-1, end_iteration=-1, weight_threshold=0.5),
        prune_scale=dict(frequency=100, start_iteration=-1, end_iteration=-1, threshold=1.0))))

PARAMS = ("positions", "density", "features_albedo", "features_specular", "rotation", "scale")


class FakeMoG:
    """The slice of threedgrut/model/model.py:45-205 that gs.py touches."""
    def __init__(self, init):
        self.device = "cpu"
        for k in PARAMS:
            setattr(self, k, torch.nn.Parameter(torch.as_tensor(init[k]).clone()))
        self.optimizer = torch.optim.Adam([dict(params=[getattr(self, k)], lr=1e-3, name=k) for k in PARAMS], eps=1e-15)
        self.scale_activation, self.scale_activation_inv = torch.exp, torch.log
        self.density_activation = torch.sigmoid
        self.density_activation_inv = lambda y: torch.log(y / (1 - y))

    @property
    def num_gaussians(self):
        return self.positions.shape[0]

    def get_positions(self):
        return self.positions

    def get_scale(self):
        return self.scale_activation(self.scale)

    def get_density(self):
        return self.density_activation(self.density)

    def clamp_density(self):
        pass


class TorchOnCpu:
    """`torch` as gs.py sees it here: zeros(device="cuda") lands on the CPU, normal() multiplies recorded unit draws."""
    def __init__(self, unit_draws):
        self._unit = unit_draws
        self.draws_used = 0

    def __getattr__(self, k):
        return getattr(torch, k)

    def zeros(self, *a, **kw):
        if kw.get("device") == "cuda":
            kw["device"] = "cpu"
        return torch.zeros(*a, **kw)

    def normal(self, mean, std):
        n = std.shape[0]
        u = self._unit[self.draws_used:self.draws_used + n]
        assert u.shape[0] == n, "not enough recorded draws"
        self.draws_used += n
        return mean + u * std


def snapshot(tag, model, strat, out):
    for k in PARAMS:
        p = getattr(model, k)
        out[f"{tag}/{k}"] = p.detach().numpy().copy()
        st = model.optimizer.state[p]
        out[f"{tag}/{k}/exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"{tag}/{k}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    out[f"{tag}/grad_norm_accum"] = strat.densify_grad_norm_accum.numpy().copy()
    out[f"{tag}/grad_norm_denom"] = strat.densify_grad_norm_denom.numpy().copy()


def main():
    @contextmanager
    def _no_range(*a, **k):
        yield
    torch.cuda.nvtx.range = _no_range
    from gen_host_golden import load_reference
    load_reference()                                     # omegaconf / tensorboard shells, threedgrut.utils.misc by path
    _shell("threedgrut.model")
    _shell("threedgrut.model.model", MixtureOfGaussians=object)
    _shell("threedgrut.utils.logger", logger=types.SimpleNamespace(info=lambda *a, **k: None))
    _shell("threedgrut.strategy")
    _load("threedgrut.strategy.base", "threedgrut/strategy/base.py")
    gs = _load("threedgrut.strategy.gs", "threedgrut/strategy/gs.py")

    rng = np.random.default_rng(20260)
    n = 240
    scene_extent = 2.5
    init = dict(
        positions=rng.uniform(-1, 1, size=(n, 3)).astype(np.float32),
        rotation=rng.normal(size=(n, 4)).astype(np.float32),                       # un-normalised, as stored
        scale=np.log(rng.uniform(0.004, 0.012, size=(n, 3))).astype(np.float32),   # around relative_size_threshold * extent = 0.025
        density=rng.normal(-1.0, 2.5, size=(n, 1)).astype(np.float32),            # logits; some below the prune threshold
        features_albedo=rng.uniform(-1, 1, size=(n, 3)).astype(np.float32),
        features_specular=rng.normal(0, 0.1, size=(n, 45)).astype(np.float32))
    init["scale"][n // 2:] += np.log(4.0).astype(np.float32)                       # second half: larger than the size threshold
    model = FakeMoG(init)
    # Adam moments as after some training (non-zero everywhere): one optimiser step on random gradients
    for k in PARAMS:
        getattr(model, k).grad = torch.as_tensor(rng.normal(size=getattr(model, k).shape).astype(np.float32) * 1e-3)
    model.optimizer.step()
    model.optimizer.zero_grad()
    unit = torch.as_tensor(rng.normal(size=(4 * n, 3)).astype(np.float32))
    proxy = TorchOnCpu(unit)
    gs.torch = proxy

    strat = gs.GSStrategy(CONF, model)
    strat.init_densification_buffer()
    out = dict(scene_extent=np.float32(scene_extent), unit_draws=unit.numpy())
    snapshot("start", model, strat, out)

    # three views: per-view position gradients (zero rows = Gaussians the view gave nothing) and sensor positions
    grads, sensors = [], []
    for v in range(3):
        g = rng.normal(size=(n, 3)).astype(np.float32) * 10.0 ** rng.uniform(-5.5, -2.5, size=(n, 1)).astype(np.float32)
        g[rng.uniform(size=n) < 0.35] = 0.0
        s = rng.uniform(-4, 4, size=3).astype(np.float32)
        grads.append(g); sensors.append(s)
        model.positions.grad = torch.as_tensor(g)
        strat.update_gradient_buffer(sensor_position=torch.as_tensor(s))
    model.positions.grad = None
    out["view_grads"], out["view_sensors"] = np.stack(grads), np.stack(sensors)
    snapshot("after_buffer", model, strat, out)

    strat.densify_gaussians(scene_extent=scene_extent)
    out["draws_used"] = np.int64(proxy.draws_used)
    snapshot("after_densify", model, strat, out)

    strat.prune_gaussians_opacity()
    snapshot("after_prune", model, strat, out)

    strat.decay_density()
    snapshot("after_decay", model, strat, out)

    strat.reset_density()
    snapshot("after_reset", model, strat, out)

    # the schedule the trainer applies these with (utils/misc.py:198-202 on configs/strategy/gs.yaml)
    misc = sys.modules["threedgrut.utils.misc"]
    steps = np.arange(0, 16001)
    d, p, r = CONF.strategy.densify, CONF.strategy.prune, CONF.strategy.reset_density
    out["schedule_densify"] = np.array([s for s in steps if misc.check_step_condition(int(s), d.start_iteration, d.end_iteration, d.frequency)], np.int32)
    out["schedule_prune"] = np.array([s for s in steps if misc.check_step_condition(int(s), p.start_iteration, p.end_iteration, p.frequency)], np.int32)
    out["schedule_reset"] = np.array([s for s in steps if misc.check_step_condition(int(s), r.start_iteration, r.end_iteration, r.frequency)], np.int32)
    out["schedule_buffer"] = np.array([s for s in steps[:20] if misc.check_step_condition(int(s), 0, d.end_iteration, 1)], np.int32)

    np.savez_compressed(os.path.join(HERE, "strategy_golden.npz"), **out)
    sizes = {k.split("/")[0]: v.shape[0] for k, v in out.items() if k.endswith("/positions")}
    print("wrote strategy_golden.npz; Gaussians per stage:", sizes, "draws used:", proxy.draws_used)


if __name__ == "__main__":
    main()
