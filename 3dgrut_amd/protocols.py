"""The `Batch` the tracer consumes — same fields and shape checks as the reference's
threedgrut/datasets/protocols.py:23-58 (the input side of the drop-in boundary)."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Batch:
    rays_ori: torch.Tensor    # [B,H,W,3] camera-space ray origins
    rays_dir: torch.Tensor    # [B,H,W,3] camera-space unit directions
    T_to_world: torch.Tensor  # [B,4,4] camera-to-world
    rgb_gt: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None
    intrinsics: Optional[list] = None
    intrinsics_OpenCVPinholeCameraModelParameters: Optional[dict] = None
    intrinsics_OpenCVFisheyeCameraModelParameters: Optional[dict] = None

    def __post_init__(self):
        b = self.T_to_world.shape[0]
        assert self.rays_ori.shape[0] == b, "rays_ori must have the same batch size"
        assert self.rays_dir.shape[0] == b, "rays_dir must have the same batch size"
        if self.rgb_gt is not None:
            assert self.rgb_gt.ndim == 4 and self.rgb_gt.shape[0] == b, "rgb_gt must be [B,H,W,3]"
        if self.mask is not None:
            assert self.mask.ndim == 4 and self.mask.shape[0] == b, "mask must be [B,H,W,1]"
        if self.intrinsics:
            assert isinstance(self.intrinsics, list) and len(self.intrinsics) == 4, "intrinsics must be [fx,fy,cx,cy]"
