"""Loss functions of the train step (reference: threedgrut/model/losses.py, trainer.py:387-450).

`fused_ssim(img1, img2, padding="valid")` has the call signature of the external CUDA package the reference
imports; here it runs the HIP kernels of csrc/gut_ssim.hip through the C ABI.  img: [B,C,H,W] (any strides:
the permuted view of the tracer's [B,H,W,3] output is consumed in place); only img1 is differentiable.
"""
import ctypes as C

import torch

from . import _capi


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        if img1.dim() != 4 or img1.shape != img2.shape:
            raise RuntimeError("[fused_ssim] expected two [B,C,H,W] tensors of equal shape")
        if not img1.is_cuda or img1.dtype != torch.float32 or img2.dtype != torch.float32:
            raise RuntimeError("[fused_ssim] expected float32 GPU tensors (there is no CPU path)")
        lib = _capi.load()
        B, Cn, H, W = img1.shape
        img2 = img2.detach()
        if img2.stride() != img1.stride():
            img2 = _match_strides(img2, img1)
        ws_bytes = lib.gut_ssim_workspace_bytes(Cn, H, W)
        ws = torch.empty((B, (ws_bytes + 3) // 4), dtype=torch.float32, device=img1.device)
        out = torch.empty((B,), dtype=torch.float32, device=img1.device)
        stream = torch.cuda.current_stream(img1.device).cuda_stream
        sb, sc, sh, sw = img1.stride()
        with torch.cuda.device(img1.device):
            for b in range(B):
                rc = lib.gut_ssim_forward(C.c_void_p(stream), Cn, H, W, sc, sh, sw, img1.data_ptr() + 4 * b * sb,
                                          img2.data_ptr() + 4 * b * sb, ws[b].data_ptr(), out[b:].data_ptr())
                if rc:
                    raise RuntimeError(f"[fused_ssim] forward failed ({rc}); images must exceed 10x10")
        ctx.save_for_backward(img1, img2, ws)
        return out.mean()

    @staticmethod
    def backward(ctx, grad_out):
        img1, img2, ws = ctx.saved_tensors
        lib = _capi.load()
        B, Cn, H, W = img1.shape
        grad = torch.empty_strided(img1.shape, img1.stride(), dtype=torch.float32, device=img1.device)
        up = (grad_out.reshape(1).to(torch.float32) / B).contiguous()
        stream = torch.cuda.current_stream(img1.device).cuda_stream
        sb, sc, sh, sw = img1.stride()
        with torch.cuda.device(img1.device):
            for b in range(B):
                rc = lib.gut_ssim_backward(C.c_void_p(stream), Cn, H, W, sc, sh, sw, img1.data_ptr() + 4 * b * sb,
                                           img2.data_ptr() + 4 * b * sb, ws[b].data_ptr(), up.data_ptr(),
                                           grad.data_ptr() + 4 * b * sb)
                if rc:
                    raise RuntimeError(f"[fused_ssim] backward failed ({rc})")
        return grad, None


def _match_strides(src, like):
    out = torch.empty_strided(like.shape, like.stride(), dtype=src.dtype, device=src.device)
    out.copy_(src)
    return out


def fused_ssim(img1, img2, padding="valid", train=True):
    if padding != "valid":
        raise RuntimeError('[fused_ssim] only padding="valid" (the reference\'s setting) is built')
    return _FusedSSIM.apply(img1, img2)


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def photometric_loss(pred_rgb, gt_rgb, lambda_l1=0.8, lambda_ssim=0.2):
    """pred/gt [B,H,W,3].  lambda_l1*L1 + lambda_ssim*(1-SSIM)  (configs/base_gs.yaml:111-119, trainer.py:425-449)."""
    s = fused_ssim(pred_rgb.permute(0, 3, 1, 2), gt_rgb.permute(0, 3, 1, 2), padding="valid")
    return lambda_l1 * l1_loss(pred_rgb, gt_rgb) + lambda_ssim * (1.0 - s)
