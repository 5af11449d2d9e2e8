"""Native train step: the same iteration as train.TrainStep (render -> 0.8*L1+0.2*(1-SSIM) -> backward -> [all-reduce]
-> Adam with the reference's learning rates), but with the Gaussian-sized work kept out of torch autograd:

    raw [N,12] --k_activate_pack--> activated [N,12] ----\
    SH  [N,48] (one tensor, no cat) ----------------------> SplatRaster.trace            (HIP)
    image-sized loss + its gradient: torch autograd on [H,W,*] tensors only (+ HIP fused SSIM)
    SplatRaster.trace_bwd(raw_parameter_grads=True) -> dRaw [N,12], dSH [N,48]            (HIP; activation chain fused
                                                                                           into the per-Gaussian epilogue)
    [RCCL all-reduce of the two gradient tensors]
    gut_adam_step on raw [N,12] and SH [N,48] with per-column learning rates              (HIP)

This removes the ~20 Gaussian-sized elementwise/cat/split passes torch makes per step in the autograd path.
The math is the same: tests compare one step of both paths (tests/test_gpu_native.py).
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _capi
from .dp import (RecordExchange, allgather_rows_, allgather_rows_async, allreduce_max_, allreduce_max_async, allreduce_mean_, allreduce_sum_async,
                 assert_replicas_identical)
from .losses import photometric_loss
from .tracer import SplatRaster, Tracer

# column layout of the raw [N,12] tensor and the reference's Adam learning rates (configs/base_gs.yaml:81-109)
RAW_COLS = dict(positions=slice(0, 3), density=slice(3, 4), rotation=slice(4, 8), scale=slice(8, 11))


def spatial_permutation(positions: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts [N,3] positions along a 3-D Morton (Z-order) curve over their bounding box, computed on the
    tensor's device (same curve as scenes.morton_order)."""
    p = positions.detach().to(torch.float64)
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).to(torch.int64).clamp_(0, (1 << bits) - 1)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True)


class NativeGaussianModel:
    """Parameters in the tracer's own layouts.  Exposes the reference's getter contract so that
    Tracer.render(model, batch) works for evaluation.

    spatial_order=True stores the Gaussians along a Morton curve (a storage decision of this trainer; the renderer's
    results do not depend on the order, except for which of two entries with bit-identical depth composites first).
    Frustum culling then leaves long runs of rows with / without tiles, which is what makes the two-pass optimiser
    (SplatRaster.optimize_rows_without_gradient) pay: HBM serves 128-byte blocks, and with randomly interleaved rows both
    passes touch most blocks.  `permutation[i]` is the index, in the order the scene was given in, of stored row i;
    NativeTrainStep.restore_spatial_order() re-sorts after densification has appended or removed rows."""

    def __init__(self, scene: dict, device="cuda", sh_degree=3, background_color="black", spatial_order=False):
        n = scene["positions"].shape[0]
        self.spatial_order = bool(spatial_order)
        self.permutation = None
        if self.spatial_order and n:
            perm = spatial_permutation(torch.as_tensor(np.ascontiguousarray(scene["positions"], np.float32))).numpy()
            scene = {k: np.asarray(v)[perm] for k, v in scene.items()}
            self.permutation = torch.as_tensor(perm, device=device)
        dens = np.clip(np.asarray(scene["density"], np.float64), 1e-6, 1 - 1e-6)
        raw = np.zeros((n, 12), np.float32)
        raw[:, 0:3] = scene["positions"]
        raw[:, 3:4] = np.log(dens / (1 - dens))
        raw[:, 4:8] = scene["rotation"]
        raw[:, 8:11] = np.log(np.asarray(scene["scale"], np.float64))
        self.raw = torch.as_tensor(raw, device=device)
        self.features = torch.as_tensor(np.ascontiguousarray(scene["features"], np.float32), device=device)
        self.n_active_features = int(sh_degree)
        self.max_n_features = 3
        self.background_color = background_color
        self.device = device

    @classmethod
    def from_tensors(cls, raw, features, sh_degree=3, background_color="black", spatial_order=False, permutation=None):
        """A model around existing parameter tensors: raw [N,12] (pos3, density logit, quat4, log-scale3, unused) and features
        [N,48], stored as they are (checkpoint resume, or a second trainer on a copy of a live state)."""
        m = cls.__new__(cls)
        if raw.dim() != 2 or raw.shape[1] != 12 or features.shape != (raw.shape[0], 48):
            raise ValueError("from_tensors: raw must be [N,12] and features [N,48]")
        m.raw, m.features = raw.contiguous(), features.contiguous()
        m.spatial_order, m.permutation = bool(spatial_order), permutation
        m.n_active_features, m.max_n_features = int(sh_degree), 3
        m.background_color, m.device = background_color, raw.device
        return m

    @property
    def num_gaussians(self):
        return self.raw.shape[0]

    @property
    def positions(self):
        return self.raw[:, 0:3]

    def get_rotation(self):
        return torch.nn.functional.normalize(self.raw[:, 4:8], dim=1)

    def get_scale(self):
        return torch.exp(self.raw[:, 8:11])

    def get_density(self):
        return torch.sigmoid(self.raw[:, 3:4])

    def get_features(self):
        return self.features

    def background(self, T_to_world, rays_d, rgb, opacity, train=False):
        if self.background_color == "white":
            rgb = rgb + (1.0 - opacity)
        elif self.background_color == "random" and train:
            rgb = rgb + torch.rand_like(rays_d) * (1.0 - opacity)
        return rgb, opacity


class NativeTrainStep:
    OVERLAP_MIN_GAUSSIANS = 1_000_000   # size from which the two-pass optimiser is on by default (see __init__)
    PROBE_FIRST, PROBE_LAST = 2, 9       # steps in which the default-on overlap is timed against the one-pass form, alternating

    def __init__(self, model: NativeGaussianModel, tracer: Tracer, scene_extent=1.0, world_size=1, selective=False,
                 betas=(0.9, 0.999), eps=1e-15, fused_sh_adam=True, rank=0, fused_loss=True, lambda_l1=0.8, lambda_ssim=0.2,
                 dp_chunks=4, dp_chunk_min_rows=1 << 20, fuse_epilogue=True, schedule=None,
                 overlap_optimizer=None, dp_exchange="sparse", dp_side_stream=True, lazy_moments=True):
        self.model = model
        self.tracer = tracer
        self.raster: SplatRaster = tracer.tracer_wrapper
        self.world_size = world_size
        self.selective = selective
        self.betas, self.eps = betas, eps
        self._lib = _capi.load()
        dev = model.raw.device
        n = model.num_gaussians
        lr12 = np.zeros(12, np.float32)
        lr12[0:3] = 1.6e-4 * scene_extent   # positions
        lr12[3] = 0.05                      # density
        lr12[4:8] = 0.001                   # rotation
        lr12[8:11] = 0.005                  # scale
        lr48 = np.full(48, 0.000125, np.float32)  # specular
        lr48[0:3] = 0.0025                  # albedo
        self.lr12, self.lr48 = lr12, lr48
        # schedule.TrainSchedule: exponentially decayed position rate + progressive SH degree (trainer.py:745-765).
        # None keeps both constant (the steady-state regime the bench measures).
        self.schedule = schedule
        if schedule is not None:
            self.lr12[0:3] = schedule.position_lr
            model.n_active_features = schedule.n_active_features
        z = lambda c: torch.zeros((n, c), dtype=torch.float32, device=dev)
        self.m12, self.v12, self.m48, self.v48 = z(12), z(12), z(48), z(48)
        self.fused = bool(fused_sh_adam)
        # Lazy moment decay (gut_hip.h: GutLazyMoments): a wave that cannot receive a gradient in a step updates its parameters
        # from moments it reads but does not write back; `wave_step[w]` holds the step up to which wave w's stored moments are
        # current and the next reader multiplies by beta^(steps missed).  m12 / v12 / m48 / v48 are therefore only current
        # after sync_moments().  Off with SelectiveAdam (no decay at all there) and with the unfused optimiser.
        self.lazy_moments = bool(lazy_moments) and self.fused and not selective
        self.LAZY_TABLE = 1024
        k = np.arange(self.LAZY_TABLE, dtype=np.float64)
        # powers of the fp32 betas the kernels multiply by (0.999f is 0.999000013: over 1000 steps the double's powers would drift 1.3e-5)
        self._pow1 = torch.as_tensor((float(np.float32(betas[0])) ** k).astype(np.float32), device=dev)
        self._pow2 = torch.as_tensor((float(np.float32(betas[1])) ** k).astype(np.float32), device=dev)
        self.wave_step = None
        self._lazy_overrun = torch.zeros(1, dtype=torch.int32, device=dev)   # GutLazyMoments.d_overrun, checked in sync_moments()
        self._step_id = 0
        self.fused_loss = bool(fused_loss)
        self.fuse_epilogue = bool(fuse_epilogue)
        # one view, fused epilogue: the Adam step of the 64-row waves that cannot receive a gradient from the view (no tile, or
        # nothing of the wave among the list entries the forward walked) runs on a side stream of the handle, under and beside
        # the compositing kernels (SplatRaster.optimize_rows_without_gradient)
        # Default: on when the model keeps its rows in spatial order — whole waves only qualify where such Gaussians are stored
        # next to each other (6 M-Gaussian bench frame, Morton-ordered rows: 3.66 -> 3.0 ms per step; with randomly ordered rows
        # hardly a wave qualifies).
        # Below ~1 M Gaussians the optimiser is a few percent of the step and the side-stream pass only costs (lego-like 300 k:
        # 877 -> 851 images/s), so the default also asks for a model of that size.
        self.overlap_optimizer = (bool(getattr(model, "spatial_order", False)) and model.num_gaussians >= self.OVERLAP_MIN_GAUSSIANS) \
            if overlap_optimizer is None else bool(overlap_optimizer)
        # When the overlap is on by DEFAULT (not forced by the caller) it is checked against the one-pass form on this machine
        # and workload: steps 2..9 alternate the two forms under event timers and the faster one (best sample) is kept (the two forms leave
        # bit-identical parameters, so the choice is invisible in the results).  Rows in an unfavourable order, a small visible
        # fraction or a box whose queues arbitrate badly can each make the one-pass form the faster one.
        self._overlap_probe = dict(on=[], off=[], done=False) if (overlap_optimizer is None and self.overlap_optimizer) else None
        self.dp_chunks, self.dp_chunk_min_rows = max(1, int(dp_chunks)), int(dp_chunk_min_rows)
        # What the ranks exchange per step (fused path): "sparse" = one 64-byte record per Gaussian a view gave a gradient to
        # (gut_compact_gradient_rows -> all-gather -> gut_scatter_gradient_records), "dense" = [N,12] all-reduce + [N,3] per
        # view all-gather, pipelined over row chunks.  Sparse moves 64 B x (Gaussians hit) per view instead of 48 + 12 B x N
        # and wins while a view hits less than about a third of the scene (a view of the bench scenes hits 4 %).
        if dp_exchange not in ("sparse", "dense"):
            raise ValueError("dp_exchange must be 'sparse' or 'dense'")
        self.dp_exchange = dp_exchange
        # sparse exchange only: the ranks MAX-reduce one byte per 64-row wave ("my forward walked a Gaussian of this wave") and the
        # waves no view walked get their zero-gradient Adam step on a side stream, under the backward compositor and the exchange
        # (gut_mark_walked_waves / gut_adam_unwalked_waves); off with SelectiveAdam, whose mask that kernel does not take
        self.dp_side_stream = bool(dp_side_stream) and not selective
        self._side = None
        self.lambda_l1, self.lambda_ssim = float(lambda_l1), float(lambda_ssim)
        self._loss_ws = None
        self._loss3 = None
        self._cam_ring = None
        self._cam_dev = None
        self._act_key = None
        self.rank = int(rank)
        self.force_exchange = os.environ.get("GUT_DP_FORCE_COLLECTIVES") == "1"  # see dp._skip
        # sparse exchange: payload sized from the previous steps' record counts, counts consumed on the device (dp.RecordExchange)
        self._exchange = RecordExchange(max(1, world_size))
        # every replica_check_every steps (and on step 0) the ranks compare checksums of parameters and moments and raise if they
        # differ (dp.assert_replicas_identical): nothing else would notice a diverged replica
        self.replica_check_every = 500
        self.post_backward_hook = None  # callable(position_grad [N,3] of THIS view, sensor_position [3]) — densification stats
        # callable() -> (norm_accum float32 [N,1], norm_denom int32 [N,1]) GPU tensors or None: the statistics of gs.py:106-115 for
        # THIS step, accumulated by the fused optimiser kernel itself (gut_set_position_gradient_statistics).  A hook that comes
        # with it (GSStrategy.attach sets both) no longer takes the step off the fused one- / two-pass optimiser: the hook is only
        # called on the paths whose optimiser kernel does not see the view's own gradient (data-parallel exchange, unfused).
        self.fused_statistics = None
        self.row_listeners = []         # callables(perm): told when reorder() re-sorts the rows (strategy statistics follow)
        self.resize_workspace()
        self.phase_timing = False   # record HIP events around the phases of step() (bench / profiling)
        self._phase_events = []

    @property
    def probe_pending(self):
        """True while the default-on optimiser overlap is still being timed against the one-pass form (steps PROBE_FIRST ..
        PROBE_LAST of single-view steps with the fused epilogue; data-parallel and hooked steps never probe)."""
        p = self._overlap_probe
        one_pass = self.fused and self.world_size <= 1 and not self.force_exchange and self.fuse_epilogue \
            and (self.post_backward_hook is None or self.fused_statistics is not None)
        return p is not None and not p["done"] and one_pass and not self.selective

    def resize_workspace(self):
        """(Re)allocate the per-step buffers for the current number of Gaussians (called after densification)."""
        n = self.model.num_gaussians
        dev = self.model.raw.device
        self._act_key = None
        if getattr(self, "lazy_moments", False):
            # (whoever changed the rows brought the moments up to date first: sync_moments)
            self.wave_step = torch.full(((n + 63) // 64,), int(getattr(self, "step_id", 0)), dtype=torch.int32, device=dev)
        self.act = torch.empty((n, 12), dtype=torch.float32, device=dev)
        self.g12 = torch.empty((n, 12), dtype=torch.float32, device=dev)
        if self.fused:
            # compact exchange: per view only dL/dRGB (12 B per Gaussian) travels; the [N,48] SH gradient is rebuilt
            # inside the fused SH-gradient + Adam kernel (csrc/gut_train.hip: k_sh_adam).  With more than one rank the
            # Gaussians are split into row chunks (multiples of 256 rows = whole optimiser workgroups) that are exchanged
            # and optimised as a pipeline; self.mrgb[c] is chunk c's gathered [world, rows_c, 3] block.
            w = max(1, self.world_size)
            nchunks = 1
            sparse = getattr(self, "dp_exchange", "dense") == "sparse"
            if (w > 1 or getattr(self, "force_exchange", False)) and n >= self.dp_chunk_min_rows and not sparse:
                nchunks = self.dp_chunks
            rows = (((n + nchunks - 1) // nchunks) + 255) // 256 * 256 if n else 0
            self.chunks = [(r0, min(n, r0 + rows)) for r0 in range(0, n, rows)] if n else []
            if sparse:
                # dense accumulators the records are scattered into: zero everywhere except where a record landed, and the
                # optimiser kernel zeroes exactly those rows again (never cleared wholesale)
                self.g12.zero_()
                self.mrgb = [torch.zeros((w, r1 - r0, 3), dtype=torch.float32, device=dev) for r0, r1 in self.chunks]
                self.records = torch.empty((max(n, 1), _capi.GRADIENT_RECORD_FLOATS), dtype=torch.float32, device=dev)
                self.rec_count = torch.zeros(1, dtype=torch.int32, device=dev)
                if getattr(self, "_exchange", None) is not None:
                    self._exchange.reset()     # the number of Gaussians changed: no history to size the payload from
                self.wave_flags = torch.zeros(((n + 63) // 64,), dtype=torch.uint8, device=dev)
                self.exchanged_records = 0     # records received in the last step, all views (diagnostics / bench)
            else:
                self.mrgb = [torch.empty((w, r1 - r0, 3), dtype=torch.float32, device=dev) for r0, r1 in self.chunks]
            self.mrgb_local = torch.empty((n, 3), dtype=torch.float32, device=dev)
            self.cams = torch.zeros((w, 3), dtype=torch.float32, device=dev)
            self.g48 = None
        else:
            self.g48 = torch.empty((n, 48), dtype=torch.float32, device=dev)

    def _lazy(self):
        """GutLazyMoments for the optimiser entry points, or None."""
        if not self.lazy_moments or self.wave_step is None:
            return None
        return _capi.GutLazyMoments(self.wave_step.data_ptr(), self._pow1.data_ptr(), self._pow2.data_ptr(), self.LAZY_TABLE,
                                    self._lazy_overrun.data_ptr())

    # The step counter and the per-wave `wave_step` of the lazy moment decay belong together: a wave's stored moments are current up
    # to wave_step[w] and every reader multiplies them by beta^(step - wave_step[w]).  So the counter cannot simply be assigned
    # (st.step_id = k on a resumed trainer left wave_step at 0 and the first step multiplied every moment by beta^1023):
    # assigning it goes through set_step(), which brings the stored moments up to date at the OLD count first and then re-bases
    # every wave on the new one; _end_of_step advances the private counter.
    @property
    def step_id(self):
        return self._step_id

    @step_id.setter
    def step_id(self, value):
        self.set_step(value)

    def set_step(self, step):
        """Set the number of optimiser steps applied so far (resume, tests).  The moments keep their values: they are brought up to
        date at the current count, then every wave is marked current at the new one."""
        step = int(step)
        if step < 0:
            raise ValueError("step must be >= 0")
        if step != self._step_id:
            self.sync_moments()
            self._step_id = step
            if self.wave_step is not None:
                self.wave_step.fill_(step)

    def state_dict(self):
        """Optimiser state for a checkpoint: the step count and the four moment tensors, CURRENT (sync_moments() first — with the
        lazy decay the stored tensors alone are stale), as clones.  Parameters are the model's (model.raw / model.features)."""
        self.sync_moments()
        return dict(step=int(self._step_id), exp_avg_raw=self.m12.clone(), exp_avg_sq_raw=self.v12.clone(),
                    exp_avg_features=self.m48.clone(), exp_avg_sq_features=self.v48.clone(), lr_raw=self.lr12.copy())

    def load_state_dict(self, state):
        """Inverse of state_dict() on a trainer of the same size: moments copied in, every wave marked current at the saved step."""
        n = self.model.num_gaussians
        for key, dst in (("exp_avg_raw", self.m12), ("exp_avg_sq_raw", self.v12), ("exp_avg_features", self.m48), ("exp_avg_sq_features", self.v48)):
            src = state[key]
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"load_state_dict: {key} has shape {tuple(src.shape)}, the trainer holds {n} Gaussians ({tuple(dst.shape)})")
            dst.copy_(src)
        if "lr_raw" in state:
            self.lr12[:] = np.asarray(state["lr_raw"], np.float32)
        self._step_id = int(state["step"])
        if self.wave_step is not None:
            self.wave_step.fill_(self._step_id)

    def sync_moments(self):
        """Bring every stored Adam moment up to the last step applied (gut_sync_moments).  Call before reading m12 / v12 / m48 /
        v48, and before anything that moves rows between 64-row waves (reorder and the strategy's row surgery do).  Raises if any
        kernel since the last call met a wave whose moments had missed more steps than the beta^k tables hold (its decay was wrong)."""
        lz = self._lazy()
        if lz is None or self.model.num_gaussians == 0 or not self.model.raw.is_cuda:
            return
        st = torch.cuda.current_stream(self.model.raw.device).cuda_stream
        with torch.cuda.device(self.model.raw.device):
            rc = self._lib.gut_sync_moments(C.c_void_p(st), self.model.num_gaussians, self.m12.data_ptr(), self.v12.data_ptr(),
                                            self.m48.data_ptr(), self.v48.data_ptr(), C.byref(lz), int(self.step_id))
        if rc:
            raise RuntimeError(f"[3dgut] sync_moments failed ({rc})")
        if int(self._lazy_overrun.item()):   # (a blocking read, on a path that runs every LAZY_TABLE / 2 steps and on row surgery)
            self._lazy_overrun.zero_()
            raise RuntimeError(f"[3dgut] lazy moment decay: a wave's stored Adam moments had missed {self.LAZY_TABLE} or more steps "
                               "(wave_step does not belong to step_id: use set_step() / load_state_dict() to resume, never a bare "
                               "assignment of wave_step or of the moments)")

    def reorder(self, perm: torch.Tensor):
        """Apply a row permutation to the parameters and the optimiser state (new row i = old row perm[i])."""
        self.sync_moments()
        m = self.model
        perm = perm.to(m.raw.device)
        m.raw = m.raw[perm].contiguous()
        m.features = m.features[perm].contiguous()
        self.m12, self.v12 = self.m12[perm].contiguous(), self.v12[perm].contiguous()
        self.m48, self.v48 = self.m48[perm].contiguous(), self.v48[perm].contiguous()
        if m.permutation is not None:
            m.permutation = m.permutation[perm]
        for fn in self.row_listeners:
            fn(perm)
        self.resize_workspace()

    def tune_placement(self, attempts=4):
        """Re-place the three [N,48] state tensors (SH parameters and their two moments) in HBM where that makes the optimiser's
        stream faster; call once after the first step or two of a run (the bench does; called between building the model and
        the first step, the compositing kernels ran 4 - 6 % slower in half of the processes — DESIGN.md §5) and again after
        densification / reorder, which re-allocate.

        On MI355X the rate at which the optimiser streams its seven tensors has two plateaus (measured on the 6 M-Gaussian bench
        scene: 5.2 and 5.9 TB/s for the same no-op pass, profiles/round3/placement_probe_*.log) decided by WHICH physical memory
        backs the three big tensors relative to one another: moving one of them to a fresh allocation flips the plateau either
        way, virtual offsets inside one allocation do not matter, and ONE arena holding all seven (contiguous or interleaved by
        64-row blocks) always lands on the slow plateau.  So: time a no-op pass of the side-stream kernel over every row (zero
        learning rates, beta = 1: every value is rewritten with itself), then give each big tensor up to `attempts` fresh
        allocations, round robin, until one move makes the pass > 5 % faster (the rates are bimodal: 1.76 - 1.78 against 1.50 -
        1.54 ms at 6 M Gaussians, so the first faster trial is the fast plateau and the search ends; with two attempts per tensor
        four processes in a row stayed on the slow plateau, with four the odds of that are small).  Values are untouched; transient
        memory is at most 3 x attempts copies of ONE [N,48] tensor (13.8 GB at 6 M Gaussians; the first version of this held
        eight copies of the whole state).  Returns the pass times in ms, first = where the state was."""
        m = self.model
        n = m.num_gaussians
        if n == 0 or not m.raw.is_cuda:
            return []
        dev = m.raw.device
        flags = torch.zeros(((n + 63) // 64,), dtype=torch.uint8, device=dev)
        zero12, zero48 = (C.c_float * 12)(), (C.c_float * 48)()
        stream = torch.cuda.current_stream(dev)

        def noop_pass():
            rc = self._lib.gut_adam_unwalked_waves(
                C.c_void_p(stream.cuda_stream), n, flags.data_ptr(), m.raw.data_ptr(), self.m12.data_ptr(), self.v12.data_ptr(),
                m.features.data_ptr(), self.m48.data_ptr(), self.v48.data_ptr(), zero12, zero48, 1.0, 1.0, self.eps, 0,
                self.act.data_ptr())
            if rc:
                raise RuntimeError(f"[3dgut] adam_unwalked_waves failed ({rc})")

        def timed(reps=3):
            noop_pass()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                noop_pass()
            e1.record(stream)
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        self.activate()   # the activation rows are part of what the pass rewrites: make them valid first
        times = [min(timed(), timed())]   # (the very first pass of a process can be a few percent slow)
        best = times[0]
        held = []         # rejected / replaced allocations stay alive until the end, so that a "fresh" allocation IS fresh
        found = False
        for _ in range(max(0, int(attempts))):       # round robin over the three tensors; the rates are bimodal, so the first
            for obj, name in ((m, "features"), (self, "m48"), (self, "v48")):   # trial that is > 5 % faster ends the search
                free_bytes, _ = torch.cuda.mem_get_info(dev)
                old = getattr(obj, name)
                if free_bytes < 2 * old.numel() * old.element_size():
                    found = True   # (no room for another copy: stop)
                    break
                setattr(obj, name, old.clone())
                t = timed()
                times.append(t)
                if t < 0.95 * best:   # the plateaus are 12 - 15 % apart; noise between passes on one placement is 1 - 3 %
                    best = t
                    held.append(old)
                    found = True
                    break
                held.append(getattr(obj, name))
                setattr(obj, name, old)
            if found:
                break
        del held
        self._act_key = None
        torch.cuda.empty_cache()
        self.placement_trials_ms = times
        return times

    def restore_spatial_order(self):
        """Re-sort the rows along the Morton curve of the CURRENT positions (after densification / pruning changed them)."""
        m = self.model
        if m.permutation is None:
            m.permutation = torch.arange(m.num_gaussians, device=m.raw.device)
        self.reorder(spatial_permutation(m.raw[:, 0:3]))
        m.spatial_order = True

    # ---- pieces ----
    def activate(self):
        raw = self.model.raw
        if self._act_key == (raw.data_ptr(), raw._version, raw.shape[0]):
            return self.act  # written by the previous step's fused Adam kernel; raw untouched since
        st = torch.cuda.current_stream(self.model.raw.device).cuda_stream
        rc = self._lib.gut_activate_pack(C.c_void_p(st), self.model.num_gaussians, self.model.raw.data_ptr(), self.act.data_ptr())
        if rc:
            raise RuntimeError(f"[3dgut] activate_pack failed ({rc})")
        return self.act

    def _adam(self, p, g, m, v, lr, vis):
        st = torch.cuda.current_stream(p.device).cuda_stream
        lr_arr = (C.c_float * len(lr))(*[float(x) for x in lr])
        rc = self._lib.gut_adam_step(C.c_void_p(st), p.shape[0], p.shape[1], p.data_ptr(), g.data_ptr(), m.data_ptr(),
                                     v.data_ptr(), lr_arr, self.betas[0], self.betas[1], self.eps,
                                     0 if self.selective else self.step_id + 1,
                                     None if vis is None else vis.data_ptr())
        if rc:
            raise RuntimeError(f"[3dgut] adam_step failed ({rc})")

    def forward(self, batch, train=True):
        """Returns (pred_rgb [1,H,W,3] leaf requiring grad, pred_opacity, aux) without involving Gaussian-sized autograd."""
        m = self.model
        act = self.activate()
        sensor, poses = Tracer.create_camera_parameters(batch)
        rgba, dist_, hits, vis = self.raster.trace(self.step_id, m.n_active_features, act, m.features, batch.rays_ori.contiguous(),
                                                  batch.rays_dir.contiguous(), None, sensor, poses.timestamps_us[0],
                                                  poses.timestamps_us[1], poses.T_world_sensors[0], poses.T_world_sensors[1])
        self._ctx = (batch, sensor, poses, rgba, dist_)
        return rgba, dist_, hits, vis

    def _sensor_position(self, batch):
        """Sensor position [3] on the device.  A host-resident batch.T_to_world (an extension of the Batch contract: the
        reference keeps it on the GPU and reads it back every step, tracer.py:353-356) goes through a small ring of
        pinned staging buffers so that no step ever blocks on the GPU."""
        t = batch.T_to_world.reshape(4, 4)[:3, 3].to(torch.float32)
        if t.is_cuda:
            return t.contiguous()
        if self._cam_ring is None:
            self._cam_ring = [torch.empty(3, dtype=torch.float32).pin_memory() for _ in range(8)]
            self._cam_dev = [torch.empty(3, dtype=torch.float32, device=self.model.raw.device) for _ in range(8)]
        k = self.step_id % len(self._cam_ring)
        self._cam_ring[k].copy_(t)
        self._cam_dev[k].copy_(self._cam_ring[k], non_blocking=True)
        return self._cam_dev[k]

    def _mark(self, evs):
        if evs is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)

    def phase_times_mean(self, reset=True):
        """Mean milliseconds per phase over the steps recorded since the last call (phase_timing must be on):
        forward (activate + trace), loss (image-sized autograd fwd+bwd), backward (trace_bwd), update (exchange + Adam)."""
        names = ("forward", "loss", "backward", "update")
        if not self._phase_events:
            return {}
        torch.cuda.synchronize(self.model.raw.device)
        acc = [0.0] * len(names)
        for evs in self._phase_events:
            for k in range(len(names)):
                acc[k] += evs[k].elapsed_time(evs[k + 1])
        n = len(self._phase_events)
        # whole steps, event to event (GPU time incl. any wait for the host): the slowest one shows a one-off stall
        spans = sorted(evs[0].elapsed_time(evs[len(names)]) for evs in self._phase_events)
        self.last_step_spans_ms = dict(min=spans[0], median=spans[len(spans) // 2], max=spans[-1])
        if os.environ.get("GUT_STEP_SPANS") == "1":
            self.last_step_spans_ms["all"] = [[round(evs[k].elapsed_time(evs[k + 1]), 3) for k in range(len(names))] for evs in self._phase_events]
        if spans[-1] > 3.0 * spans[len(spans) // 2]:   # a one-off stall: say where (step index within the timed region, phase)
            worst = max(range(n), key=lambda i: self._phase_events[i][0].elapsed_time(self._phase_events[i][len(names)]))
            evs = self._phase_events[worst]
            self.last_step_spans_ms["slowest"] = dict(index=worst, **{names[k]: evs[k].elapsed_time(evs[k + 1]) for k in range(len(names))})
        if reset:
            self._phase_events = []
        return {names[k]: acc[k] / n for k in range(len(names))}

    def step(self, batch):
        try:
            return self._step(batch)
        except Exception as err:
            # The side-stream optimiser pass may already be running for this iteration (it is queued right behind the forward):
            # finish the step for every other row with a zero gradient instead of leaving the parameters half advanced and the
            # handle refusing the next forward (gut_optimize_finish_without_gradient); then let the error through.
            if getattr(self, "_early_queued", False):
                self._early_queued = False
                try:
                    self.raster.finish_optimizer_step_without_gradient()
                except Exception as finish_err:
                    # the handle is half applied and cannot be repaired from here (a HIP error, most likely): say both
                    raise RuntimeError(f"[3dgut] the optimiser step begun on the side stream could not be finished after "
                                       f"{type(err).__name__}: {err}; the trainer's state is not usable: {finish_err}") from err
                self._act_key = None
                self._probe_evs = None
                # the iteration WAS applied — every row took its step, with a zero gradient — so it goes through the same
                # bookkeeping as any other (learning-rate / SH-degree schedule, sync cadence of the lazy moments, step count)
                self._end_of_step(None)
            raise

    def _step(self, batch):
        m = self.model
        if self.world_size > 1 and self.replica_check_every and not getattr(self, "_replicas_checked_at_start", False):
            # BEFORE the first update of this trainer (`step_id % every == 0` alone fires after it): ranks that were built from
            # different scenes, seeds or checkpoints are caught while nothing has been applied yet
            self._replicas_checked_at_start = True
            assert_replicas_identical([self.model.raw, self.model.features, self.m12, self.v12, self.m48, self.v48], self.world_size,
                                      what=f"before the first step (step counter {self.step_id})")
        evs = [] if self.phase_timing else None
        self._mark(evs)
        one_pass = self.fused and self.world_size <= 1 and not self.force_exchange and self.fuse_epilogue \
            and (self.post_backward_hook is None or self.fused_statistics is not None)
        use_overlap, probe_evs = self.overlap_optimizer, None
        probe = self._overlap_probe
        if probe is not None and one_pass and not self.selective:
            if self.PROBE_FIRST <= self.step_id <= self.PROBE_LAST:      # probing steps: off, on, off, on, ...
                use_overlap = (self.step_id % 2) == 1
                probe_evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                probe_evs[0].record()
                probe["on" if use_overlap else "off"].append(probe_evs)
            elif self.step_id > self.PROBE_LAST and not probe["done"] and all(e[1].query() for e in probe["on"] + probe["off"]):
                # the FASTEST sample of each form: the first steps of a run also pay for allocations, a binning overflow, the creation
                # of the side stream (the two-sample means of rounds 2 decided wrongly on the survey-C3 stand-in: 3.92 vs 3.86 ms
                # measured in steps 1-4 against 3.34 vs 3.67 ms in steady state)
                t_on = min((a.elapsed_time(b) for a, b in probe["on"]), default=float("inf"))
                t_off = min((a.elapsed_time(b) for a, b in probe["off"]), default=float("inf"))
                probe.update(done=True, ms_on=t_on, ms_off=t_off)
                if probe["on"] and probe["off"]:   # (a trainer whose step counter starts beyond the probing steps keeps the default)
                    self.overlap_optimizer = use_overlap = t_on <= t_off
        early = one_pass and use_overlap and not self.selective
        self._probe_evs = probe_evs
        rgba, dist_, hits, vis = self.forward(batch)
        if early:
            self.raster.optimize_rows_without_gradient(m.raw, self.m12, self.v12, m.features, self.m48, self.v48, self.lr12,
                                                       self.lr48, self.betas, self.eps, self.step_id + 1, self.act, lazy=self._lazy())
            self._early_queued = True
        self._mark(evs)
        gt = batch.rgb_gt
        if self.fused_loss and m.background_color in ("black", "white") and gt.dtype == torch.float32 and gt.is_contiguous() \
                and gt.numel() == rgba.shape[0] * rgba.shape[1] * 3:
            # loss value and d(loss)/d(rgba) in two HIP launches (csrc/gut_ssim.hip: gut_photometric_loss)
            H, W = rgba.shape[0], rgba.shape[1]
            need = self._lib.gut_photometric_workspace_bytes(H, W)
            if self._loss_ws is None or self._loss_ws.numel() * 4 < need:
                self._loss_ws = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=rgba.device)
            # three fresh floats every step (the caching allocator, no kernel): the loss returned below is a VIEW of them — a
            # `.clone()` of a buffer kept across steps was a 7 us copy kernel between the loss and the backward
            self._loss3 = torch.empty((3,), dtype=torch.float32, device=rgba.device)
            rgba_grad = torch.empty_like(rgba)
            st = torch.cuda.current_stream(rgba.device).cuda_stream
            rc = self._lib.gut_photometric_loss(C.c_void_p(st), H, W, rgba.data_ptr(), gt.data_ptr(),
                                                1.0 if m.background_color == "white" else 0.0, self.lambda_l1, self.lambda_ssim,
                                                self._loss_ws.data_ptr(), self._loss3.data_ptr(), rgba_grad.data_ptr())
            if rc:
                raise RuntimeError(f"[3dgut] photometric_loss failed ({rc})")
            loss = self._loss3[0]
            pred_rgb = rgba[..., :3].unsqueeze(0)
            if m.background_color == "white":
                pred_rgb = pred_rgb + (1.0 - rgba[..., 3:].unsqueeze(0))
        else:
            rgba_leaf = rgba.detach().requires_grad_(True)
            pred_rgb = rgba_leaf[..., :3].unsqueeze(0)
            pred_opacity = rgba_leaf[..., 3:].unsqueeze(0)
            pred_rgb, pred_opacity = m.background(batch.T_to_world, batch.rays_dir, pred_rgb, pred_opacity, True)
            loss = photometric_loss(pred_rgb, gt, self.lambda_l1, self.lambda_ssim)
            loss.backward()  # image-sized autograd only
            rgba_grad = rgba_leaf.grad
        self._mark(evs)
        _, sensor, poses, _, _ = self._ctx
        bwd_args = (self.step_id, m.n_active_features, self.act, m.features, batch.rays_ori.contiguous(),
                    batch.rays_dir.contiguous(), None, sensor, poses.timestamps_us[0], poses.timestamps_us[1],
                    poses.T_world_sensors[0], poses.T_world_sensors[1], rgba, rgba_grad, dist_, None)
        vmask = None
        if self.fused:
            w = max(1, self.world_size)
            exchange = w > 1 or self.force_exchange
            if one_pass:
                # one view, nobody else needs the per-Gaussian gradients: K8's epilogue, the SH-gradient rebuild and Adam run
                # as ONE pass over the Gaussians, straight from the renderer's gradient rows (gut_optimize_after_bwd)
                self.raster.trace_bwd(*bwd_args, skip_epilogue=True)
                self._mark(evs)
                vmask = vis.reshape(-1) if self.selective else None
                stats = self.fused_statistics() if self.fused_statistics is not None else None
                if stats is not None:
                    self.raster.set_position_gradient_statistics(*stats)
                # camera position None: the library uses the sensor position of its cached forward (the very floats K1 evaluated the
                # colours from; an 18 us host-to-device copy sat here, between K7 and the pass over the walked waves, until round 4)
                self.raster.optimize_after_bwd(m.n_active_features, None, m.raw, self.m12, self.v12, m.features,
                                               self.m48, self.v48, self.lr12, self.lr48, self.betas, self.eps,
                                               0 if self.selective else self.step_id + 1, vmask, self.act, lazy=self._lazy())
                self._act_key = (m.raw.data_ptr(), m.raw._version, m.raw.shape[0])
                self._mark(evs)
                self._end_of_step(evs)
                return loss.detach(), dict(pred_rgb=pred_rgb.detach(), mog_visibility=vis, hits_count=hits)
            if self.dp_exchange == "sparse":
                self._sparse_exchange_and_update(batch, bwd_args, vis, evs, w, exchange)
                self._mark(evs)
                self._end_of_step(evs)
                return loss.detach(), dict(pred_rgb=pred_rgb.detach(), mog_visibility=vis, hits_count=hits)
            # this view's compact radiance gradient: directly view 0 of the gathered layout when there is nothing to gather
            local_mrgb = self.mrgb_local if exchange else self.mrgb[0][0]
            self.raster.trace_bwd(*bwd_args, raw_parameter_grads=True, compact_radiance_grads=True, out=(self.g12, local_mrgb))
            self._mark(evs)
            cam_local = self._sensor_position(batch)
            if self.post_backward_hook is not None:  # per-view statistics, before the exchange (strategy/gs.py:106-115)
                self.post_backward_hook(self.g12[:, 0:3], cam_local)
            works = []
            if exchange:
                # Chunk-pipelined exchange: the collectives of chunk c+1 run on RCCL's stream while the optimiser kernel
                # of chunk c runs on the compute stream (the optimiser is HBM-bound, the exchange xGMI-bound).
                allgather_rows_(self.cams, cam_local, w)
                if self.selective:
                    allreduce_max_(vis, w)
                for (r0, r1), gathered in zip(self.chunks, self.mrgb):
                    works.append((allreduce_sum_async(self.g12[r0:r1], w),
                                  allgather_rows_async(gathered, self.mrgb_local[r0:r1], w)))
            else:
                self.cams[0].copy_(cam_local)
            if self.selective:
                vmask = vis.reshape(-1)
            st = torch.cuda.current_stream(m.raw.device).cuda_stream
            f32p = C.POINTER(C.c_float)
            for k, ((r0, r1), gathered) in enumerate(zip(self.chunks, self.mrgb)):
                if works:
                    works[k][0].wait()
                    works[k][1].wait()
                lz = self._lazy()
                if lz is not None:   # chunks are multiples of 256 rows: whole waves
                    lz = _capi.GutLazyMoments(self.wave_step.data_ptr() + 4 * (r0 // 64), self._pow1.data_ptr(), self._pow2.data_ptr(), self.LAZY_TABLE)
                rc = self._lib.gut_sh_adam_step_ex(
                    C.c_void_p(st), r1 - r0, m.n_active_features, w, self.cams.data_ptr(), gathered.data_ptr(),
                    self.g12.data_ptr() + 48 * r0, 1.0 / w, m.raw.data_ptr() + 48 * r0, self.m12.data_ptr() + 48 * r0,
                    self.v12.data_ptr() + 48 * r0, m.features.data_ptr() + 192 * r0, self.m48.data_ptr() + 192 * r0,
                    self.v48.data_ptr() + 192 * r0, self.lr12.ctypes.data_as(f32p), self.lr48.ctypes.data_as(f32p),
                    self.betas[0], self.betas[1], self.eps, 0 if self.selective else self.step_id + 1,
                    None if vmask is None else vmask.data_ptr() + 4 * r0, self.act.data_ptr() + 48 * r0, r1 - r0, 0, None,
                    None if lz is None else C.byref(lz))
                if rc:
                    raise RuntimeError(f"[3dgut] sh_adam_step failed ({rc})")
            # any in-place torch edit of raw (densification, MCMC noise, ...) bumps _version and forces a fresh activation
            self._act_key = (m.raw.data_ptr(), m.raw._version, m.raw.shape[0])
        else:
            self.raster.trace_bwd(*bwd_args, raw_parameter_grads=True, out=(self.g12, self.g48))
            self._mark(evs)
            if self.post_backward_hook is not None:
                self.post_backward_hook(self.g12[:, 0:3], batch.T_to_world.reshape(4, 4)[:3, 3].to(torch.float32))
            if self.world_size > 1 or self.force_exchange:
                allreduce_mean_([self.g48, self.g12], self.world_size)
                if self.selective:
                    allreduce_max_(vis, self.world_size)
            vmask = vis.reshape(-1) if self.selective else None
            self._adam(m.raw, self.g12, self.m12, self.v12, self.lr12, vmask)
            self._adam(m.features, self.g48, self.m48, self.v48, self.lr48, vmask)
        self._mark(evs)
        self._end_of_step(evs)
        return loss.detach(), dict(pred_rgb=pred_rgb.detach(), mog_visibility=vis, hits_count=hits)

    def _sparse_exchange_and_update(self, batch, bwd_args, vis, evs, w, exchange):
        """Backward + sparse gradient exchange + optimiser of a data-parallel step (dp_exchange == "sparse")."""
        m = self.model
        n = m.num_gaussians
        f32p = C.POINTER(C.c_float)
        main = torch.cuda.current_stream(m.raw.device)
        side_on = self.dp_side_stream and n > 0
        if side_on:
            # one byte per wave: did THIS view's forward walk a Gaussian of it?  MAX over the ranks while the backward runs.
            rc = self._lib.gut_mark_walked_waves(self.raster._handle, C.c_void_p(main.cuda_stream), self.wave_flags.data_ptr())
            _capi.check(rc, "mark_walked_waves")
            flags_ready = torch.cuda.Event()
            flags_ready.record(main)               # BEFORE the backward is queued: the side stream must not wait for it
            flags_work = allreduce_max_async(self.wave_flags, w) if exchange else None
        self.raster.trace_bwd(*bwd_args, skip_epilogue=True)
        if side_on:
            if self._side is None:
                self._side = torch.cuda.Stream(device=m.raw.device)
            self._side.wait_event(flags_ready)
            with torch.cuda.stream(self._side):
                if flags_work is not None:
                    flags_work.wait()
                lz = self._lazy()
                rc = self._lib.gut_adam_unwalked_waves_ex(
                    C.c_void_p(self._side.cuda_stream), n, self.wave_flags.data_ptr(), m.raw.data_ptr(), self.m12.data_ptr(),
                    self.v12.data_ptr(), m.features.data_ptr(), self.m48.data_ptr(), self.v48.data_ptr(),
                    self.lr12.ctypes.data_as(f32p), self.lr48.ctypes.data_as(f32p), self.betas[0], self.betas[1], self.eps,
                    self.step_id + 1, self.act.data_ptr(), None if lz is None else C.byref(lz))
                if rc:
                    raise RuntimeError(f"[3dgut] adam_unwalked_waves failed ({rc})")
            if flags_work is not None:
                flags_work.wait()                  # the main stream's optimiser call reads the reduced flags too
        self.raster.compact_gradient_rows(self.act, self.records, self.rec_count)
        self._mark(evs)
        cam_local = self._sensor_position(batch)
        if exchange:
            allgather_rows_(self.cams, cam_local, w)
            if self.selective:
                allreduce_max_(vis, w)
        else:
            self.cams[0].copy_(cam_local)
        # queued without reading anything back: counts all-gather, payload all-gather at the capacity carried from earlier steps,
        # one scatter per view (rank order on every rank: identical summation order, identical replicas) with the view's record
        # count taken on the device
        ex = self._exchange
        gathered, counts_dev, cap = ex.start(self.records, self.rec_count)
        st = torch.cuda.current_stream(m.raw.device).cuda_stream
        slabs = self.mrgb[0]
        for v in range(gathered.shape[0]):
            if cap:
                rc = self._lib.gut_scatter_gradient_records_dev(C.c_void_p(st), gathered[v].data_ptr(), counts_dev.data_ptr() + 4 * v, cap, n,
                                                                self.g12.data_ptr(), slabs[v].data_ptr())
                if rc:
                    raise RuntimeError(f"[3dgut] scatter_gradient_records failed ({rc})")
        counts = ex.host_counts()         # everything above is queued: the GPU keeps working while the host waits for 4 W bytes
        tail = ex.tail(self.records)      # a rank had more records than the capacity assumed: the rest, same order on every rank
        if tail is not None:
            for v, c in enumerate(tail[1]):
                if c:
                    rc = self._lib.gut_scatter_gradient_records(C.c_void_p(st), tail[0][v].data_ptr(), c, n, self.g12.data_ptr(),
                                                                slabs[v].data_ptr())
                    if rc:
                        raise RuntimeError(f"[3dgut] scatter_gradient_records failed ({rc})")
        self.exchanged_records = int(sum(counts))
        self.exchanged_bytes_per_rank = int(ex.payload_bytes_per_rank)
        if self.post_backward_hook is not None:  # per-view statistics of THIS rank's view (strategy/gs.py:106-115)
            r = self.rank if len(counts) > 1 else 0
            mine = gathered[r, :min(counts[r], cap)]
            if tail is not None and tail[1][r]:
                mine = torch.cat([mine, tail[0][r, :tail[1][r]]])
            pg = torch.zeros((n, 3), dtype=torch.float32, device=m.raw.device)
            pg[mine[:, 11].contiguous().view(torch.int32).long()] = mine[:, 0:3]
            self.post_backward_hook(pg, cam_local)
        vmask = vis.reshape(-1) if self.selective else None
        rc = self._lib.gut_sh_adam_step_ex(
            C.c_void_p(st), n, m.n_active_features, w, self.cams.data_ptr(), slabs.data_ptr(), self.g12.data_ptr(), 1.0 / w,
            m.raw.data_ptr(), self.m12.data_ptr(), self.v12.data_ptr(), m.features.data_ptr(), self.m48.data_ptr(),
            self.v48.data_ptr(), self.lr12.ctypes.data_as(f32p), self.lr48.ctypes.data_as(f32p), self.betas[0], self.betas[1],
            self.eps, 0 if self.selective else self.step_id + 1, None if vmask is None else vmask.data_ptr(), self.act.data_ptr(),
            n, _capi.ADAM_CLEAR_CONSUMED_GRADS, self.wave_flags.data_ptr() if side_on else None,
            None if self._lazy() is None else C.byref(self._lazy()))
        if rc:
            raise RuntimeError(f"[3dgut] sh_adam_step failed ({rc})")
        if side_on:
            main.wait_stream(self._side)           # the next forward reads every row
        self._act_key = (m.raw.data_ptr(), m.raw._version, m.raw.shape[0])

    def _end_of_step(self, evs):
        self._early_queued = False
        if self.world_size > 1 and self.replica_check_every and self.step_id % self.replica_check_every == 0:
            assert_replicas_identical([self.model.raw, self.model.features, self.m12, self.v12, self.m48, self.v48], self.world_size,
                                      what=f"after step {self.step_id}")
        if getattr(self, "_probe_evs", None) is not None:
            self._probe_evs[1].record()
            self._probe_evs = None
        if evs is not None:
            self._phase_events.append(evs)
        if self.schedule is not None:   # scheduler_step(g) + SH-degree increase, after the optimiser (trainer.py:756-765)
            lr, deg = self.schedule.after_optimizer_step(self.step_id)
            self.lr12[0:3] = lr         # passed by value to the optimiser kernels of the next step
            self.model.n_active_features = deg
        self._step_id += 1
        if self.lazy_moments and self._step_id % (self.LAZY_TABLE // 2) == 0:
            self.sync_moments()   # long before any wave's missed steps run off the end of the beta^k tables

    def render(self, batch, train=False):
        return self.tracer.render(self.model, batch, train=train, frame_id=self.step_id)
