"""MI355X-native 3DGUT differentiable Gaussian renderer (hot path of OrangeEarth15/3dgrut).

The directory name starts with a digit, so import it with importlib:

    import importlib; gut = importlib.import_module("3dgrut_amd"); tracer = gut.Tracer(conf)

or through the drop-in alias package at the repository root: `import threedgut_tracer`.
"""
from .protocols import Batch  # noqa: F401
from .tracer import (  # noqa: F401
    CameraModelParameters,
    ShutterType,
    SplatRaster,
    Tracer,
    fromOpenCVFisheyeCameraModelParameters,
    fromOpenCVPinholeCameraModelParameters,
)

__all__ = ["Tracer", "SplatRaster", "ShutterType", "CameraModelParameters", "Batch",
           "fromOpenCVPinholeCameraModelParameters", "fromOpenCVFisheyeCameraModelParameters"]
