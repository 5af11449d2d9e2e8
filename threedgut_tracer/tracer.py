"""`threedgut_tracer.tracer` module path kept for callers that import it directly
(reference: threedgrut/model/model.py:184-185 does `import threedgut_tracer` and uses `.Tracer`)."""
import importlib as _il

_t = _il.import_module("3dgrut_amd.tracer")
Tracer = _t.Tracer
SplatRaster = _t.SplatRaster
SensorPose3D = _t.SensorPose3D
__all__ = ["Tracer", "SplatRaster", "SensorPose3D"]
