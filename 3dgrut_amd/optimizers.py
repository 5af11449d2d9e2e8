"""`SelectiveAdam` with the reference's surface (threedgrut/optimizers/__init__.py:46-131): a `torch.optim.Adam` subclass whose
`step(visibility)` updates only the Gaussians the last view saw — rows with visibility 0 keep parameters AND moments, no bias
correction — one parameter tensor per group.  The update is the HIP kernel behind `gut_selective_adam` (the reference JIT-builds
`lib_optimizers_cc` from optimizers.cu for it).  No CPU fallback: parameters must live on the GPU.

    opt = SelectiveAdam(model.param_groups(extent), eps=1e-15, betas=(0.9, 0.999))
    out = tracer.render(model, batch, train=True); loss.backward()
    opt.step(out["mog_visibility"]); opt.zero_grad()
"""
import ctypes as C

import torch

from . import _capi


class SelectiveAdam(torch.optim.Adam):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-08):
        super().__init__(params=params, lr=lr, eps=eps, betas=betas)
        self._lib = _capi.load()

    @torch.no_grad()
    def step(self, visibility):
        vis = visibility.bool().squeeze().contiguous()
        for group in self.param_groups:
            lr, eps = group["lr"], group["eps"]
            beta1, beta2 = group["betas"]
            assert len(group["params"]) == 1, "More than one tensor in group is not supported"
            param = group["params"][0]
            if param.grad is None:
                continue
            if not param.is_cuda or param.dtype != torch.float32:
                raise RuntimeError("[3dgut] SelectiveAdam: parameters must be float32 GPU tensors (there is no CPU path)")
            state = self.state[param]
            if len(state) == 0:   # lazy state initialisation, as in the reference
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            exp_avg, exp_avg_sq = state["exp_avg"], state["exp_avg_sq"]
            n = param.shape[0]
            if n == 0:
                continue
            if vis.numel() != n:
                raise RuntimeError(f"[3dgut] SelectiveAdam: visibility has {vis.numel()} entries for {n} rows")
            if not (param.is_contiguous() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()):
                raise RuntimeError("[3dgut] SelectiveAdam: parameters and moments must be contiguous")   # (the reference's .contiguous() would update a copy)
            grad = param.grad.contiguous()
            stream = torch.cuda.current_stream(param.device).cuda_stream
            with torch.cuda.device(param.device):
                rc = self._lib.gut_selective_adam(C.c_void_p(stream), n, param.numel() // n, param.data_ptr(), grad.data_ptr(),
                                                  exp_avg.data_ptr(), exp_avg_sq.data_ptr(), vis.data_ptr(), float(lr), float(beta1),
                                                  float(beta2), float(eps))
            if rc:
                raise RuntimeError(f"[3dgut] selective_adam failed ({rc})")
