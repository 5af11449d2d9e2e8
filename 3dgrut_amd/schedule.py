"""Per-step schedules of the reference's training loop that touch the hot path's inputs: the exponentially decayed
position learning rate and the progressive SH degree.

Reference behaviour (pinned by tests/golden/host_golden.npz, generated from the reference's own Python):
  * `exponential_scheduler(lr_init, lr_final, max_steps)` — threedgrut/utils/misc.py:89-96: log-linear interpolation,
    `t = clip(step / max_steps, 0, 1)`; the positions' `lr_init` / `lr_final` are multiplied by the scene extent
    (threedgrut/model/model.py:528-538); every other group has `type: skip` (configs/base_gs.yaml:98-109).
  * order inside one iteration `g` (threedgrut/trainer.py:705-767): render(frame_id=g) -> loss -> backward ->
    optimizer.step() -> `scheduler_step(g)` (the new rate applies from iteration g+1 on) -> if progressive training and
    `check_step_condition(g, 0, 1e6, increase_frequency)`: `n_active_features = min(max, n + increase_step)`
    (model.py:566-567); defaults init 0, max 3, every 1000 steps (configs/base_gs.yaml:65-70).
"""
import math


def exponential_scheduler(lr_init, lr_final, max_steps=1000000):
    log_i, log_f = math.log(lr_init), math.log(lr_final)

    def helper(step):
        t = min(max(step / max_steps, 0.0), 1.0)
        return math.exp(log_i * (1 - t) + log_f * t)

    return helper


def check_step_condition(step, start, end, freq):
    """threedgrut/utils/misc.py:198-202."""
    return bool((start >= 0 and step > start) and (step < end or end == -1) and step % freq == 0)


class TrainSchedule:
    """State of the two schedules; `after_optimizer_step(g)` is called once per iteration, after Adam, with the index of
    the iteration that just ran — exactly where the reference calls `scheduler_step` / `increase_num_active_features`."""

    def __init__(self, scene_extent=1.0, lr_init=0.00016, lr_final=0.0000016, max_steps=30000, init_n_features=0,
                 max_n_features=3, increase_frequency=1000, increase_step=1):
        self._sched = exponential_scheduler(lr_init * scene_extent, lr_final * scene_extent, max_steps)
        self.position_lr = lr_init * scene_extent       # optimizer.params.positions.lr * extent (model.py:518-520)
        self.n_active_features = int(init_n_features)
        self.max_n_features = int(max_n_features)
        self.progressive = self.n_active_features < self.max_n_features   # model.py:171-179
        self.increase_frequency = int(increase_frequency)
        self.increase_step = int(increase_step)

    def after_optimizer_step(self, global_step):
        self.position_lr = self._sched(global_step)
        if self.progressive and check_step_condition(global_step, 0, 1e6, self.increase_frequency):
            self.n_active_features = min(self.max_n_features, self.n_active_features + self.increase_step)
        return self.position_lr, self.n_active_features
