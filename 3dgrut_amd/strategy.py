"""Densification / pruning strategies on the native parameter layout ("next" row N3 of SURVEY §8f).

Restates, on the tensors of native.NativeTrainStep (raw [N,12], features [N,48] and their Adam moments), what the
reference does with per-name nn.Parameters and optimizer-state surgery:
  * GSStrategy   — threedgrut/strategy/gs.py:60-306 (gradient-norm buffer, clone, split, opacity prune, density
                   decay / reset), base.py:52-83 (parameter + optimizer-state update rule: moments of new rows are 0);
  * MCMCStrategy — threedgrut/strategy/mcmc.py:76-197 (relocate dead Gaussians, add new ones, perturb), with the
                   relocation kernel of src/gaussian_mcmc.cu as HIP (gut_mcmc_relocation).
All of it is torch tensor surgery that runs every 100-3000 steps, far off the per-step hot path; the only per-step
piece is update_gradient_buffer.  Data parallel: the per-view norms are accumulated BEFORE the gradient exchange
(gs.py:106-115 works on per-view gradients) and SUM-reduced at densification time; random splits draw from a
generator seeded identically on every rank so replicas stay bit-identical.
"""
import ctypes as C
import math

import torch

from . import _capi

POS, DNS, ROT, SCL = slice(0, 3), slice(3, 4), slice(4, 8), slice(8, 11)


def _quat_to_rotmat(q):
    """rows = rotationT rows?  The reference uses quaternion_to_so3 (wxyz, normalised) -> R with columns = axes."""
    q = torch.nn.functional.normalize(q, dim=1)
    w, x, y, z = q.unbind(1)
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)


def multinomial_sample(weights, n, generator=None):
    """`n` indices drawn with replacement, probability proportional to `weights` (flat, non-negative).
    torch.multinomial refuses more than 2^24 categories; the reference falls back to numpy on the host there
    (threedgrut/utils/misc.py:164-196).  Here larger inputs are sampled on the device by inverting the cumulative sum
    (float64, one searchsorted), which draws from the same distribution without leaving the GPU."""
    assert weights.dim() == 1, "multinomial_sample expects a flat tensor"
    if weights.shape[0] <= 2 ** 24:
        return torch.multinomial(weights, n, replacement=True, generator=generator)
    cdf = torch.cumsum(weights.to(torch.float64), 0)
    u = torch.rand(n, dtype=torch.float64, device=weights.device, generator=generator) * cdf[-1]
    return torch.searchsorted(cdf, u, right=True).clamp_(max=weights.shape[0] - 1)


class _StateOps:
    """Row surgery on (raw, features) + Adam moments of a NativeTrainStep."""

    def __init__(self, stepper):
        self.s = stepper
        self.m = stepper.model

    @property
    def n(self):
        return self.m.raw.shape[0]

    def _sync(self):
        sync = getattr(self.s, "sync_moments", None)   # lazily decayed moments are brought up to date before rows move between waves
        if sync is not None:
            sync()

    def _apply(self, fn_param, fn_moment):
        s, m = self.s, self.m
        self._sync()
        m.raw = fn_param(m.raw)
        m.features = fn_param(m.features)
        s.m12, s.v12 = fn_moment(s.m12), fn_moment(s.v12)
        s.m48, s.v48 = fn_moment(s.m48), fn_moment(s.v48)
        s.resize_workspace()

    def keep(self, mask):
        idx = mask.nonzero().squeeze(1)   # once, not once per tensor (p[mask] finds the indices — and synchronises — every time)
        self._apply(lambda p: p.index_select(0, idx), lambda v: v.index_select(0, idx))
        perm = getattr(self.m, "permutation", None)
        if perm is not None:   # stored row i came from row permutation[i] of the scene as given; rows added later get -1
            self.m.permutation = perm.index_select(0, idx.to(perm.device))

    def append(self, raw_new, feat_new):
        s, m = self.s, self.m
        self._sync()
        k = raw_new.shape[0]
        m.raw = torch.cat([m.raw, raw_new]).contiguous()
        m.features = torch.cat([m.features, feat_new]).contiguous()
        z = lambda v: torch.cat([v, torch.zeros((k, v.shape[1]), dtype=v.dtype, device=v.device)]).contiguous()
        s.m12, s.v12, s.m48, s.v48 = z(s.m12), z(s.v12), z(s.m48), z(s.v48)
        perm = getattr(m, "permutation", None)
        if perm is not None:
            m.permutation = torch.cat([perm, torch.full((k,), -1, dtype=perm.dtype, device=perm.device)])
        s.resize_workspace()


# configs/strategy/gs.yaml: (start_iteration, end_iteration, frequency) of each operation; -1 start = never
GS_SCHEDULE = dict(densify=(500, 15000, 300), prune=(500, 15000, 100), reset_density=(0, 15000, 3000), density_decay=(-1, -1, 50))


class GSStrategy:
    """threedgrut/strategy/gs.py:26-306 on the native layout.  Defaults are configs/strategy/gs.yaml's.

    Live use with a NativeTrainStep (what trainer.py:741-760 does with its strategy object):
        gs = GSStrategy(stepper); gs.attach()          # per-view statistics hook + row bookkeeping
        for step ...: stepper.step(batch); gs.post_optimizer_step(step, scene_extent)
    attach() makes the stepper hand this view's position gradient to update_gradient_buffer after every backward
    (post_backward, gs.py:60-73) and keeps the statistics rows aligned when the stepper re-sorts its rows."""

    def __init__(self, stepper, clone_grad_threshold=0.0002, split_grad_threshold=0.0002, relative_size_threshold=0.01,
                 split_n_gaussians=2, prune_density_threshold=0.005, new_max_density=0.01, density_decay_gamma=0.99, seed=0,
                 schedule=None):
        self.ops = _StateOps(stepper)
        self.clone_thr, self.split_thr = clone_grad_threshold, split_grad_threshold
        self.rel_size, self.split_n = relative_size_threshold, split_n_gaussians
        self.prune_thr, self.new_max_density, self.decay_gamma = prune_density_threshold, new_max_density, density_decay_gamma
        self.seed = seed
        self.schedule = dict(GS_SCHEDULE if schedule is None else schedule)
        # split(): standard-normal draws [k * split_n, 3] for the children's offsets; None = a generator seeded with
        # (seed, step), identical on every data-parallel rank.  (The reference draws torch.normal on the CUDA generator, gs.py:145.)
        self.unit_normal_fn = None
        self.reset_buffers()

    # ---- trainer callbacks (gs.py:60-104) ----
    def attach(self):
        s = self.ops.s
        s.post_backward_hook = self._post_backward
        if hasattr(s, "fused_statistics"):
            s.fused_statistics = self._fused_statistics   # one view, fused optimiser: the optimiser kernel accumulates them itself
        listeners = getattr(s, "row_listeners", None)
        if listeners is not None and self._rows_reordered not in listeners:
            listeners.append(self._rows_reordered)
        return self

    def detach(self):
        s = self.ops.s
        if getattr(s, "post_backward_hook", None) == self._post_backward:
            s.post_backward_hook = None
        if getattr(s, "fused_statistics", None) == self._fused_statistics:
            s.fused_statistics = None

    def _fused_statistics(self):
        """The buffers the trainer's fused optimiser kernel accumulates this step's statistics into, or None when this step has none
        (same schedule as _post_backward)."""
        from .schedule import check_step_condition
        if not check_step_condition(int(getattr(self.ops.s, "step_id", 1)), 0, self.schedule["densify"][1], 1):
            return None
        if not self.grad_norm_accum.is_cuda or self.grad_norm_accum.shape[0] != self.ops.n:
            raise RuntimeError("[3dgut] densification buffers and the model differ in their number of rows")
        return self.grad_norm_accum, self.grad_norm_denom

    def _post_backward(self, position_grad, sensor_position):
        """gs.py:60-66: the buffer is updated for 0 < step < densify.end_iteration."""
        from .schedule import check_step_condition
        if check_step_condition(int(getattr(self.ops.s, "step_id", 1)), 0, self.schedule["densify"][1], 1):
            self.update_gradient_buffer(position_grad, sensor_position)

    def _rows_reordered(self, perm):
        perm = perm.to(self.grad_norm_accum.device)
        self.grad_norm_accum, self.grad_norm_denom = self.grad_norm_accum[perm], self.grad_norm_denom[perm]

    def post_optimizer_step(self, step, scene_extent, world=1):
        """gs.py:75-104 with utils/misc.check_step_condition.  Returns True when the number or order of rows changed."""
        from .schedule import check_step_condition
        sc, updated, appended = self.schedule, False, False
        if check_step_condition(step, *sc["densify"]):
            self.densify(scene_extent, step, world)
            updated = appended = True
        if check_step_condition(step, *sc["prune"]):
            self.prune_opacity()
            updated = True
        if check_step_condition(step, *sc["density_decay"]):
            self.decay_density()
        if check_step_condition(step, *sc["reset_density"]):
            self.reset_density()
        if step >= sc["densify"][1]:
            self.detach()   # update_gradient_buffer only runs up to densify.end_iteration (gs.py:64)
        if appended and getattr(self.ops.m, "spatial_order", False):
            self.ops.s.restore_spatial_order()   # (pruning alone keeps the rows in their order along the curve)
        return updated

    def reset_buffers(self):
        dev = self.ops.m.raw.device
        self.grad_norm_accum = torch.zeros((self.ops.n, 1), dtype=torch.float32, device=dev)
        self.grad_norm_denom = torch.zeros((self.ops.n, 1), dtype=torch.int32, device=dev)

    @torch.no_grad()
    def update_gradient_buffer(self, position_grad, sensor_position):
        """gs.py:106-115; position_grad = this view's dL/dpositions [N,3] (before any cross-rank exchange)."""
        if position_grad.is_cuda and position_grad.dtype == torch.float32 and position_grad.stride(1) == 1:
            # one kernel (gut_position_gradient_statistics) instead of ten torch launches and a boolean-index synchronisation
            raw = self.ops.m.raw
            cam = sensor_position.to(device=raw.device, dtype=torch.float32).reshape(-1)[:3].contiguous()
            st = torch.cuda.current_stream(raw.device).cuda_stream
            with torch.cuda.device(raw.device):
                rc = _capi.load().gut_position_gradient_statistics(C.c_void_p(st), raw.shape[0], position_grad.data_ptr(), position_grad.stride(0),
                                                                   raw.data_ptr(), raw.stride(0), cam.data_ptr(),
                                                                   self.grad_norm_accum.data_ptr(), self.grad_norm_denom.data_ptr())
            if rc:
                raise RuntimeError(f"[3dgut] position_gradient_statistics failed ({rc})")
            return
        mask = (position_grad != 0).any(dim=1)
        dist = (self.ops.m.raw[:, POS][mask] - sensor_position).norm(dim=1, keepdim=True)
        self.grad_norm_accum[mask] += torch.norm(position_grad[mask] * dist, dim=-1, keepdim=True) / 2
        self.grad_norm_denom[mask] += 1

    @torch.no_grad()
    def densify(self, scene_extent, step=0, world=1):
        if world > 1:  # per-view statistics are additive over ranks
            import torch.distributed as dist
            dist.all_reduce(self.grad_norm_accum)
            dist.all_reduce(self.grad_norm_denom)
        g = self.grad_norm_accum / self.grad_norm_denom
        g[g.isnan()] = 0.0
        g = g.squeeze(1)
        self.clone(g, scene_extent)
        self.split(g, scene_extent, step)

    @torch.no_grad()
    def clone(self, grad_norm, scene_extent):
        m = self.ops.m
        scale_max = torch.exp(m.raw[:, SCL]).max(dim=1).values
        mask = (grad_norm >= self.clone_thr) & (scale_max <= self.rel_size * scene_extent)
        self.ops.append(m.raw[mask], m.features[mask])
        self.reset_buffers()
        return int(mask.sum())

    @torch.no_grad()
    def split(self, grad_norm, scene_extent, step=0):
        m = self.ops.m
        n0 = self.ops.n
        padded = torch.zeros(n0, device=m.raw.device)
        padded[: grad_norm.shape[0]] = grad_norm  # cloned rows appended after the statistics were taken get 0
        scale = torch.exp(m.raw[:, SCL])
        mask = (padded >= self.split_thr) & (scale.max(dim=1).values > self.rel_size * scene_extent)
        k = int(mask.sum())
        if k:
            stds = scale[mask].repeat(self.split_n, 1)
            if self.unit_normal_fn is not None:
                unit = self.unit_normal_fn(stds.shape).to(stds.device)
            else:
                gen = torch.Generator(device=m.raw.device).manual_seed(self.seed * 1_000_003 + step)
                unit = torch.randn(stds.shape, generator=gen, device=stds.device)
            samples = unit * stds
            rots = _quat_to_rotmat(m.raw[:, ROT][mask]).repeat(self.split_n, 1, 1)
            offsets = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1)
            raw_new = m.raw[mask].repeat(self.split_n, 1)
            raw_new[:, POS] += offsets
            raw_new[:, SCL] = torch.log(torch.exp(raw_new[:, SCL]) / (0.8 * self.split_n))
            feat_new = m.features[mask].repeat(self.split_n, 1)
            self.ops.keep(~mask)
            self.ops.append(raw_new, feat_new)
        self.reset_buffers()
        return k

    @torch.no_grad()
    def prune_opacity(self):
        keep = torch.sigmoid(self.ops.m.raw[:, 3]) >= self.prune_thr
        self.ops.keep(keep)
        self.grad_norm_accum, self.grad_norm_denom = self.grad_norm_accum[keep], self.grad_norm_denom[keep]
        return int((~keep).sum())

    @torch.no_grad()
    def decay_density(self):
        d = torch.sigmoid(self.ops.m.raw[:, 3:4].contiguous()) * self.decay_gamma   # contiguous [N,1], as the reference's parameter
        self.ops.m.raw[:, 3:4] = torch.log(d / (1 - d))

    @torch.no_grad()
    def reset_density(self):
        cap = math.log(self.new_max_density / (1 - self.new_max_density))
        self.ops.m.raw[:, 3].clamp_(max=cap)
        self.ops.s.m12[:, 3] = 0
        self.ops.s.v12[:, 3] = 0


# configs/strategy/mcmc.yaml: (start_iteration, end_iteration, frequency) of each operation
MCMC_SCHEDULE = dict(relocate=(500, 25000, 100), add=(500, 25000, 100), perturb=(0, 27500, 1))


class MCMCStrategy:
    """threedgrut/strategy/mcmc.py:48-197 on the native trainer's state.  Three things come from outside the host logic and can be
    replaced (the reference pins of tests/test_cpu_reference_pins.py feed the recorded ones): `sample_fn(weights, n, step)` —
    the multinomial draw (default: torch.multinomial / the device-side inverse-CDF draw above, seeded per step);
    `relocation_fn(opacities, scales, ratios)` — the relocation kernel (default: gut_mcmc_relocation, HIP, no CPU path);
    `unit_normal_fn(shape, step)` — the standard-normal draws of the perturbation."""

    def __init__(self, stepper, opacity_threshold=0.005, binom_n_max=51, max_n_gaussians=1_000_000, noise_lr=5e5, seed=0, schedule=None):
        self.ops = _StateOps(stepper)
        self.opacity_threshold, self.n_max, self.max_n, self.noise_lr, self.seed = opacity_threshold, binom_n_max, max_n_gaussians, noise_lr, seed
        self.schedule = dict(MCMC_SCHEDULE, **(schedule or {}))
        self.sample_fn = self.relocation_fn = self.unit_normal_fn = None
        b = torch.zeros((binom_n_max, binom_n_max), dtype=torch.float32)
        for n in range(binom_n_max):
            for k in range(n + 1):
                b[n, k] = math.comb(n, k)
        self.binoms = b.to(self.ops.m.raw.device)

    def post_optimizer_step(self, step, position_lr):
        """mcmc.py:76-90: relocate, add, perturb — each on the iterations its (start, end, frequency) selects."""
        from .schedule import check_step_condition
        done = []
        if check_step_condition(step, *self.schedule["relocate"]):
            self.relocate(step); done.append("relocate")
        if check_step_condition(step, *self.schedule["add"]):
            self.add_new(step); done.append("add")
        if check_step_condition(step, *self.schedule["perturb"]):
            self.perturb(position_lr, step); done.append("perturb")
        return done

    def _relocation(self, dens, scales, ratios):
        if self.relocation_fn is not None:
            return self.relocation_fn(dens, scales, ratios)
        if not dens.is_cuda:
            raise RuntimeError("[3dgut] MCMC relocation: the kernel is HIP only (there is no CPU path)")
        lib = _capi.load()
        new_d, new_s = torch.empty_like(dens), torch.empty_like(scales)
        st = torch.cuda.current_stream(dens.device).cuda_stream
        rc = lib.gut_mcmc_relocation(C.c_void_p(st), dens.shape[0], dens.data_ptr(), scales.data_ptr(), ratios.data_ptr(),
                                     self.binoms.data_ptr(), self.n_max, new_d.data_ptr(), new_s.data_ptr())
        if rc:
            raise RuntimeError(f"[3dgut] mcmc_relocation failed ({rc})")
        return new_d, new_s

    @torch.no_grad()
    def sample_new(self, num, valid_idx=None, step=0):
        """mcmc.py:166-197"""
        m = self.ops.m
        dens = torch.sigmoid(m.raw[:, 3].contiguous())   # (contiguous: the CPU's vectorised and strided sigmoid / exp differ in the last bit)
        scales = torch.exp(m.raw[:, SCL].contiguous())
        if valid_idx is None:
            valid_idx = torch.arange(dens.shape[0], device=dens.device)
        if self.sample_fn is not None:
            drawn = self.sample_fn(dens[valid_idx], num, step)
        else:
            drawn = multinomial_sample(dens[valid_idx], num, torch.Generator(device=dens.device).manual_seed(self.seed * 1_000_003 + step))
        sampled = valid_idx[drawn]
        ratios = (torch.bincount(sampled, minlength=dens.shape[0])[sampled] + 1).clamp_(min=1, max=self.n_max).int()
        new_d, new_s = self._relocation(dens[sampled].contiguous(), scales[sampled].contiguous(), ratios.contiguous())
        new_d = new_d.clamp(max=1.0 - torch.finfo(torch.float32).eps, min=self.opacity_threshold)
        return sampled, torch.log(new_d / (1 - new_d)), torch.log(new_s)

    @torch.no_grad()
    def relocate(self, step=0):
        m, s = self.ops.m, self.ops.s
        dens = torch.sigmoid(m.raw[:, 3].contiguous())
        dead = torch.where(dens <= self.opacity_threshold)[0]
        alive = torch.where(dens > self.opacity_threshold)[0]
        if dead.numel() and alive.numel():
            sampled, nd, ns = self.sample_new(dead.numel(), alive, step)
            m.raw[sampled, 3] = nd
            m.raw[sampled, 8:11] = ns
            m.raw[dead] = m.raw[sampled]
            m.features[dead] = m.features[sampled]
            for v in (s.m12, s.v12, s.m48, s.v48):
                v[sampled] = 0
        return int(dead.numel())

    @torch.no_grad()
    def add_new(self, step=0):
        cur = self.ops.n
        add = max(0, min(self.max_n, int(1.05 * cur)) - cur)
        if add:
            sampled, nd, ns = self.sample_new(add, None, step)
            m = self.ops.m
            m.raw[sampled, 3] = nd
            m.raw[sampled, 8:11] = ns
            self.ops.append(m.raw[sampled], m.features[sampled])
        return add

    @torch.no_grad()
    def perturb(self, position_lr, step=0):
        m = self.ops.m
        if m.raw.is_cuda:
            # one kernel over the raw rows (gut_mcmc_perturb): the reference's [N,3,3] covariance matmuls take 250 ms at 6 M Gaussians
            unit = None
            if self.unit_normal_fn is not None:
                unit = self.unit_normal_fn(m.raw[:, POS].shape, step).to(device=m.raw.device, dtype=torch.float32).contiguous()
            act = getattr(self.ops.s, "act", None)
            if act is not None and (act.shape[0] != m.raw.shape[0] or not act.is_cuda):
                act = None
            st = torch.cuda.current_stream(m.raw.device).cuda_stream
            with torch.cuda.device(m.raw.device):
                rc = _capi.load().gut_mcmc_perturb(C.c_void_p(st), m.raw.shape[0], m.raw.data_ptr(), None if act is None else act.data_ptr(),
                                                   float(self.noise_lr * position_lr), int(self.seed), int(step),
                                                   None if unit is None else unit.data_ptr())
            if rc:
                raise RuntimeError(f"[3dgut] mcmc_perturb failed ({rc})")
            return
        R = _quat_to_rotmat(m.raw[:, ROT])
        S = torch.diag_embed(torch.exp(m.raw[:, SCL].contiguous()))
        cov = R @ S @ S.transpose(1, 2) @ R.transpose(1, 2)
        dens = torch.sigmoid(m.raw[:, 3:4].contiguous())
        if self.unit_normal_fn is not None:
            unit = self.unit_normal_fn(m.raw[:, POS].shape, step)
        else:
            gen = torch.Generator(device=m.raw.device).manual_seed(self.seed * 1_000_003 + step + 7)
            unit = torch.randn(m.raw[:, POS].shape, generator=gen, device=m.raw.device)
        noise = unit * (1 / (1 + torch.exp(-100 * ((1 - dens) - 0.995)))) * self.noise_lr * position_lr
        m.raw[:, POS] += torch.bmm(cov, noise.unsqueeze(-1)).squeeze(-1)
