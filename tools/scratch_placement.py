"""Dev probe (GPU): whose physical placement do the compositing kernels' two speeds belong to?  
Builds the bench trainer in the order that draws the slow state in about half of the processes (placement tuning BEFORE the first
step), then moves one buffer at a time to a fresh allocation — the handle's scratch through GUT_OPT_DEBUG_REPLACE_SCRATCH, the
trainer's tensors by cloning — and prints the library's own K6 / K7 timers over eight steps after each move."""
import importlib, os, sys, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes"); cams = importlib.import_module("3dgrut_amd.cameras")
native = importlib.import_module("3dgrut_amd.native")
dev = torch.device("cuda", 0)
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
scene = getattr(scenes, fn)(**kw)
model = native.NativeGaussianModel(scene, device=dev, spatial_order=True)
st = native.NativeTrainStep(model, gut.Tracer({"render": {"enable_kernel_timings": True}}), scene_extent=extent)
bench.synthetic_optimizer_state(st)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev)
T = dict(ro=torch.as_tensor(ro, device=dev), rd=torch.as_tensor(rd, device=dev), gt=torch.rand((1, H, W, 3), device=dev))
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
def batch(i):
    return gut.Batch(rays_ori=T["ro"], rays_dir=T["rd"], T_to_world=torch.as_tensor(c2ws[i % 8])[None], rgb_gt=T["gt"], intrinsics_OpenCVPinholeCameraModelParameters=K)
if "--tune-late" not in sys.argv:
    st.tune_placement()
for i in range(2): st.step(batch(i))
if "--tune-late" in sys.argv:
    st.tune_placement()
for i in range(2, 14): st.step(batch(i))
raster = st.raster
def measure(label):
    for i in range(3): st.step(batch(i))
    raster.kernel_times_mean()
    for i in range(8): st.step(batch(i))
    t, _ = raster.kernel_times_mean()
    print(f"{label:34s} K6 {t['render']:.3f}  K7 {t['render_bwd']:.3f}  K1 {t['project']:.3f}  late {t['optimizer']:.3f}", flush=True)
    return t["render"], t["render_bwd"]
base = measure("as built")
names = ["tiles_count", "tiles_offset", "proj_pos", "conic_opacity", "extent", "depth", "feat", "gradient rows", "scan temp", "keys unsorted",
         "keys grouped", "ids unsorted", "ids grouped", "sort temp", "ids ordered", "tile ranges", "trav fwd", "trav bwd", "tile order", "tile ordered"]
for idx in (6, 7, 14, 12, 10, 3, 2, 4, 5, 0, 15, 16, 17, 18, 19):
    raster.debug_replace_scratch(idx)
    measure("scratch: " + names[idx])
held = []
for name in ("act",):
    old = getattr(st, name); held.append(old); setattr(st, name, old.clone()); measure("trainer: " + name)
for name in ("ro", "rd", "gt"):
    held.append(T[name]); T[name] = T[name].clone(); measure("bench tensor: " + name)
held.append(model.raw); model.raw = model.raw.clone(); st._act_key = None; measure("trainer: raw")
# torch's per-step image tensors: push the caching allocator onto other blocks
spacers = [torch.empty(int(s * 2 ** 20), dtype=torch.uint8, device=dev) for s in (4, 16, 16, 24, 64)]
measure("per-step tensors (after spacers)")
del spacers; gc.collect(); torch.cuda.empty_cache()
measure("per-step tensors (spacers freed)")
measure("again")
