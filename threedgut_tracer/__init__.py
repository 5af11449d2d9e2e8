"""Drop-in alias: `import threedgut_tracer; threedgut_tracer.Tracer(conf)` resolves to the MI355X-native
implementation in 3dgrut_amd/ (same surface as the reference's threedgut_tracer/__init__.py +
tracer.py).  See INTEGRATION.md."""
import importlib as _il

_impl = _il.import_module("3dgrut_amd")
Tracer = _impl.Tracer
SplatRaster = _impl.SplatRaster
__all__ = ["Tracer", "SplatRaster"]
