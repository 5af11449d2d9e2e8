"""Dev probe (GPU): what MCMCStrategy.post_optimizer_step adds to a native train step on the bicycle stand-in."""
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
mc = strategy.MCMCStrategy(st, max_n_gaussians=6_000_000)
for with_mc in (False, True, False, True):
    for i in range(14): st.step(batch(i))
    torch.cuda.synchronize(); gc.collect()
    t0 = time.perf_counter()
    for i in range(16):
        st.step(batch(i))
        if with_mc:
            mc.post_optimizer_step(st.step_id + 10_000_000 if False else 7, 1.6e-4)   # step 7: perturb only
    torch.cuda.synchronize()
    print(f"MCMC perturb every step: {with_mc}  {(time.perf_counter() - t0) / 16 * 1e3:.2f} ms/step", flush=True)
