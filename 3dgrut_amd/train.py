"""Train-step harness for the 3DGUT path: the call sequence of the reference's hot loop
(threedgrut/trainer.py:705-778) with nothing but the renderer swapped —

    outputs = model(gpu_batch, train=True)      -> Tracer.render (HIP)
    loss    = 0.8 * L1 + 0.2 * (1 - SSIM)       (configs/base_gs.yaml:111-119, trainer.py:425-450)
    loss.backward()                              -> Tracer._Autograd.backward (HIP)
    [data parallel: all-reduce of the Gaussian gradients over RCCL]
    optimizer.step(); optimizer.zero_grad()     (Adam, eps 1e-15, reference learning rates)

The reference's SSIM is the external CUDA-only `fused_ssim` package (model/losses.py:17-33,
requirements.txt:23); here it is the HIP kernel pair of csrc/gut_ssim.hip (losses.fused_ssim, same call
signature, `padding="valid"`).  `ssim()` below is the plain-torch fp32/fp64 reference the tests check it against.
Per-view data parallelism: one process per GPU, full replica, rank r renders view
step*world+r, gradients are summed with an RCCL all-reduce and divided by the world size.
"""
import math

import torch
import torch.nn.functional as F

from .dp import allreduce_mean_
from .losses import photometric_loss
from .optimizers import SelectiveAdam


def _gauss_window(size=11, sigma=1.5, device="cpu", dtype=torch.float32):
    x = torch.arange(size, dtype=dtype, device=device) - (size - 1) / 2
    g = torch.exp(-(x * x) / (2 * sigma * sigma))
    return g / g.sum()


def ssim(img1, img2, window=None):
    """img: [B,3,H,W] in [0,1]; separable Gaussian window, valid padding; returns the mean SSIM."""
    c = img1.shape[1]
    if window is None:
        window = _gauss_window(device=img1.device, dtype=img1.dtype)
    wh = window.view(1, 1, 1, -1).expand(c, 1, 1, -1)
    wv = window.view(1, 1, -1, 1).expand(c, 1, -1, 1)

    def blur(x):
        return F.conv2d(F.conv2d(x, wh, groups=c), wv, groups=c)

    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = blur(img1 * img1) - mu1_sq
    s2 = blur(img2 * img2) - mu2_sq
    s12 = blur(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean()


def photometric_loss_torch(pred_rgb, gt_rgb, lambda_l1=0.8, lambda_ssim=0.2, window=None):
    """pred/gt: [B,H,W,3].  0.8*L1 + 0.2*(1-SSIM) (trainer.py:449)."""
    l1 = (pred_rgb - gt_rgb).abs().mean()
    s = ssim(pred_rgb.permute(0, 3, 1, 2), gt_rgb.permute(0, 3, 1, 2), window)
    return lambda_l1 * l1 + lambda_ssim * (1.0 - s)


class TrainStep:
    def __init__(self, model, tracer, scene_extent=1.0, world_size=1, fused_adam=True, schedule=None, optimizer_type="adam"):
        """optimizer_type: configs/base_gs.yaml:82 `optimizer.type` — "adam" (torch.optim.Adam) or "selective_adam" (model.py:512-513:
        the reference's SelectiveAdam plugin, here optimizers.SelectiveAdam over `gut_selective_adam`; the step then passes the view's
        visibility, trainer.py:747-749)."""
        self.model = model
        self.tracer = tracer
        self.world_size = world_size
        if optimizer_type == "adam":
            kw = dict(eps=1e-15)
            if fused_adam and next(model.parameters()).is_cuda:
                kw["fused"] = True
            self.optimizer = torch.optim.Adam(model.param_groups(scene_extent), **kw)
        elif optimizer_type == "selective_adam":
            if world_size > 1:
                raise ValueError("selective_adam masks by ONE view's visibility: it has no data-parallel form (the reference trains single-GPU)")
            self.optimizer = SelectiveAdam(model.param_groups(scene_extent), eps=1e-15)
        else:
            raise ValueError(f"Unknown optimizer type: {optimizer_type}")
        self.window = _gauss_window(device=next(model.parameters()).device)
        self.step_id = 0
        self.schedule = schedule   # schedule.TrainSchedule or None (constant rates, fixed SH degree)
        if schedule is not None:
            self._set_schedule_state(schedule.position_lr, schedule.n_active_features)

    def _set_schedule_state(self, lr, deg):
        for group in self.optimizer.param_groups:   # model.scheduler_step, model.py:540-545
            if group["name"] == "positions":
                group["lr"] = lr
        self.model.n_active_features = deg

    def render(self, batch, train=True):
        return self.tracer.render(self.model, batch, train=train, frame_id=self.step_id)

    phase_timing = False    # bench: HIP events between the phases of step() (phase_times_mean)

    def _mark(self, evs):
        if evs is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)

    def phase_times_mean(self):
        """Mean ms per phase over the steps since the last call: `render` = Tracer.render as the trainer calls it (the model's
        activations and accessors + the plugin's forward), `loss` = the loss forward, `backward` = loss.backward() (loss backward,
        the plugin's backward, the activations' backward, gradient accumulation), `optimizer` = optimizer.step() + zero_grad()."""
        evs_all = getattr(self, "_phase_events", [])
        if not evs_all:
            return {}
        torch.cuda.synchronize()
        names = ("render", "loss", "backward", "optimizer")
        acc = [sum(e[k].elapsed_time(e[k + 1]) for e in evs_all) / len(evs_all) for k in range(4)]
        self._phase_events = []
        return dict(zip(names, acc))

    def step(self, batch):
        evs = [] if self.phase_timing else None
        self._mark(evs)
        out = self.render(batch, train=True)
        self._mark(evs)
        loss = photometric_loss(out["pred_rgb"], batch.rgb_gt)
        self._mark(evs)
        loss.backward()
        if self.world_size > 1:
            self.allreduce_gradients()
        self._mark(evs)
        if isinstance(self.optimizer, SelectiveAdam):
            assert out["mog_visibility"].shape == self.model.density.shape
            self.optimizer.step(out["mog_visibility"])
        else:
            self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        self._mark(evs)
        if evs is not None:
            self.__dict__.setdefault("_phase_events", []).append(evs)
        if self.schedule is not None:
            self._set_schedule_state(*self.schedule.after_optimizer_step(self.step_id))
        self.step_id += 1
        return loss, out

    def allreduce_gradients(self):
        """SUM over ranks then divide: the loss of a step is the mean over the views of all ranks (dp.py)."""
        for p in self.model.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        allreduce_mean_([p.grad for p in self.model.parameters()], self.world_size)
