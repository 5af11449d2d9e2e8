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


@pytest.mark.parametrize("background", ["black", "white"])
@pytest.mark.parametrize("hw", [(40, 56), (37, 53), (822, 1237)])
def test_fused_photometric_loss_and_gradient(background, hw):
    """gut_photometric_loss (loss + d/d(rgba) without autograd) vs fp64 torch autograd of
    0.8*L1 + 0.2*(1-SSIM) on rgb + background*(1-alpha).  Tolerance: |loss diff| <= 5e-6, gradient rel-L2 <= 1e-4."""
    import ctypes as C
    capi = importlib.import_module("3dgrut_amd._capi")
    lib = capi.load()
    H, W = hw
    g = torch.Generator().manual_seed(3)
    rgba = torch.rand((H, W, 4), generator=g)
    gt = torch.rand((H, W, 3), generator=g)
    bg = 1.0 if background == "white" else 0.0
    x = rgba.cuda().contiguous(); y = gt.cuda().contiguous()
    ws = torch.empty(((lib.gut_photometric_workspace_bytes(H, W) + 3) // 4,), dtype=torch.float32, device="cuda")
    out3 = torch.empty(3, dtype=torch.float32, device="cuda")
    grad = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.gut_photometric_loss(C.c_void_p(st), H, W, x.data_ptr(), y.data_ptr(), bg, 0.8, 0.2, ws.data_ptr(), out3.data_ptr(),
                                  grad.data_ptr())
    assert rc == 0
    r64 = rgba.double().requires_grad_(True)
    img = r64[..., :3] + bg * (1.0 - r64[..., 3:])
    ref = train.photometric_loss_torch(img.unsqueeze(0), gt.double().unsqueeze(0), window=train._gauss_window(dtype=torch.float64))
    ref.backward()
    o = out3.cpu().double()
    assert abs(float(o[0]) - float(ref)) <= 5e-6
    assert abs(float(o[1]) - float((img - gt.double()).abs().mean())) <= 2e-6
    got = grad.cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - r64.grad).norm() / r64.grad.norm())
    assert err <= 1e-4, err
