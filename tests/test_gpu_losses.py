"""HIP fused SSIM vs a plain PyTorch fp64 reference of the same op (11x11 Gaussian window, valid padding).
Tolerance: |mean SSIM diff| <= 2e-6, gradient rel-L2 <= 1e-4 (fp32 kernel vs fp64 reference)."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
losses = importlib.import_module("3dgrut_amd.losses")
train = importlib.import_module("3dgrut_amd.train")


@pytest.mark.parametrize("shape", [(1, 3, 64, 80), (1, 3, 37, 53), (2, 3, 48, 48), (1, 1, 822, 1237)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc_view"])
def test_fused_ssim_matches_torch(shape, layout):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    a = torch.rand(shape, generator=g)
    b = (a + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    if layout == "nhwc_view":
        x = a.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
        y = b.permute(0, 2, 3, 1).contiguous().cuda()
        s = losses.fused_ssim(x.permute(0, 3, 1, 2), y.permute(0, 3, 1, 2))
    else:
        x = a.clone().cuda().requires_grad_(True)
        y = b.cuda()
        s = losses.fused_ssim(x, y)
    (3.0 * s).backward()
    a64 = a.double().requires_grad_(True)
    ref = train.ssim(a64, b.double(), window=train._gauss_window(dtype=torch.float64))
    (3.0 * ref).backward()
    assert abs(float(s) - float(ref)) <= 2e-6
    gx = x.grad.permute(0, 3, 1, 2) if layout == "nhwc_view" else x.grad
    err = float((gx.cpu().double() - a64.grad).norm() / a64.grad.norm())
    assert err <= 1e-4, err


def test_photometric_loss_value():
    g = torch.Generator().manual_seed(1)
    pred = torch.rand((1, 40, 56, 3), generator=g).cuda().requires_grad_(True)
    gt = torch.rand((1, 40, 56, 3), generator=g).cuda()
    loss = losses.photometric_loss(pred, gt)
    ref = train.photometric_loss_torch(pred.detach().cpu().double(), gt.cpu().double(), window=train._gauss_window(dtype=torch.float64))
    assert abs(float(loss) - float(ref)) <= 5e-6
    loss.backward()
    assert torch.isfinite(pred.grad).all()
