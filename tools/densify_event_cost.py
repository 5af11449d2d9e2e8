"""Dev probe (GPU): what the GSStrategy events (prune / densify / Morton re-sort) cost on the bicycle stand-in."""
import importlib, os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes"); cams = importlib.import_module("3dgrut_amd.cameras")
native = importlib.import_module("3dgrut_amd.native"); strategy = importlib.import_module("3dgrut_amd.strategy")
dev = torch.device("cuda", 0)
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
scene = getattr(scenes, fn)(**kw)
model = native.NativeGaussianModel(scene, device=dev, spatial_order=True)
st = native.NativeTrainStep(model, gut.Tracer({"render": {}}), scene_extent=extent)
bench.synthetic_optimizer_state(st)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev)
ro_t, rd_t = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
gt = torch.rand((1, H, W, 3), device=dev)
def batch(i):
    return gut.Batch(rays_ori=ro_t, rays_dir=rd_t, T_to_world=torch.as_tensor(c2ws[i % 8])[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K)
gs = strategy.GSStrategy(st).attach()
for i in range(12): st.step(batch(i))
def timed(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) * 1e3:.1f} ms  (N = {model.num_gaussians})", flush=True)
    return r
model.raw[::50, 3] = -9.0      # 2 % below the pruning threshold
timed("prune_opacity (2 % of the rows)", gs.prune_opacity)
timed("restore_spatial_order", st.restore_spatial_order)
timed("post_optimizer_step(600): densify + prune + re-sort", lambda: gs.post_optimizer_step(600, extent))
timed("post_optimizer_step(700): prune (+ re-sort?)", lambda: gs.post_optimizer_step(700, extent))
model.raw[::40, 3] = -9.0
timed("prune_opacity again (2.5 % of the rows)", gs.prune_opacity)
timed("restore_spatial_order again", st.restore_spatial_order)
for i in range(3): st.step(batch(i))
timed("post_optimizer_step(900): densify + prune + re-sort, second time", lambda: gs.post_optimizer_step(900, extent))
for i in range(4): st.step(batch(i))
timed("4 more steps", lambda: [st.step(batch(i)) for i in range(4)])
print("peak memory GB", torch.cuda.max_memory_allocated() / 1e9)
