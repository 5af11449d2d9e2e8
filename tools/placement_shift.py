"""Dev probe (GPU): is the fast / slow placement of the trainer state a function of the virtual address?  Carves the seven state
tensors out of ONE arena at a sliding 2 MB-granular shift and times the no-op optimiser pass for every shift."""
import importlib, os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
native = importlib.import_module("3dgrut_amd.native")
dev = "cuda:0"
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
sc = getattr(scenes, fn)(**kw)
nm = native.NativeGaussianModel(sc, device=dev, spatial_order=True)
ts = native.NativeTrainStep(nm, gut.Tracer({"render": {}}), scene_extent=extent, overlap_optimizer=True)
bench.synthetic_optimizer_state(ts)
ts.activate()
lib = ts._lib
n = nm.num_gaussians
flags0 = torch.zeros(((n + 63) // 64,), dtype=torch.uint8, device=dev)
zero12, zero48 = (C.c_float * 12)(), (C.c_float * 48)()
names = ((nm, "raw"), (nm, "features"), (ts, "m12"), (ts, "v12"), (ts, "m48"), (ts, "v48"), (ts, "act"))
def noop_ms(reps=3):
    st = torch.cuda.current_stream().cuda_stream
    def once():
        assert lib.gut_adam_unwalked_waves(C.c_void_p(st), n, flags0.data_ptr(), nm.raw.data_ptr(), ts.m12.data_ptr(), ts.v12.data_ptr(),
                                           nm.features.data_ptr(), ts.m48.data_ptr(), ts.v48.data_ptr(), zero12, zero48, 1.0, 1.0, ts.eps, 0,
                                           ts.act.data_ptr()) == 0
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
MB2 = 2 << 20
sizes = [getattr(o, k).numel() * 4 for o, k in names]
rounded = [(s + MB2 - 1) // MB2 * MB2 for s in sizes]
gap = int(sys.argv[1]) if len(sys.argv) > 1 else 0     # extra 2 MB units between consecutive tensors
arena = torch.empty(sum(rounded) + (64 + 8 * gap) * MB2, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + MB2 - 1) // MB2 * MB2 - arena.data_ptr()
print("arena 2MB index", hex((arena.data_ptr() + base) >> 21), "initial (separate allocations)", round(noop_ms(), 3), flush=True)
orig = [getattr(o, k) for o, k in names]
for shift in range(0, 24):
    off = base + shift * MB2
    for (o, k), src, sz, r in zip(names, orig, sizes, rounded):
        view = arena[off:off + sz].view(torch.float32).view(src.shape)
        view.copy_(src)
        setattr(o, k, view)
        off += r + gap * MB2
    print("shift", shift, "m48 idx mod 6:", (ts.m48.data_ptr() >> 21) % 6, "raw idx mod 6:", (nm.raw.data_ptr() >> 21) % 6, "no-op ms", round(noop_ms(), 3), flush=True)
