"""Dev tool: a few forward(+backward) renders of the bench scene, for rocprofv3 runs."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
cams = importlib.import_module("3dgrut_amd.cameras"); model_mod = importlib.import_module("3dgrut_amd.model")
dev = "cuda:0"
W, H, fx = 1237, 822, 1040.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bwd = len(sys.argv) > 3 and sys.argv[3] == "bwd"
step = len(sys.argv) > 3 and sys.argv[3] == "step"   # full native train steps (incl. the one-pass optimiser kernel)
ro, rd = cams.pinhole_rays(W, H, fx, fx); K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
sc = scenes.scene_outdoor_like(n=n, seed=2)
model = None if step else model_mod.GaussianModel(sc, device=dev)
# FWD_ONCE_KERNEL_DEGREE=n: render.particle_kernel_degree (the kGeneral compositors) instead of the default 2
tr = gut.Tracer({"render": {"enable_kernel_timings": True, "particle_kernel_degree": int(os.environ.get("FWD_ONCE_KERNEL_DEGREE", "2"))}})
b = gut.Batch(rays_ori=torch.as_tensor(ro, device=dev), rays_dir=torch.as_tensor(rd, device=dev),
              T_to_world=torch.as_tensor(cams.orbit_c2w(4.5, 7.0, 12.0), device=dev)[None], intrinsics_OpenCVPinholeCameraModelParameters=K)
if step:
    # the bench's own views and target, cycled as the bench cycles them, with the two-pass optimiser forced on (no probe): the
    # LAST step's dispatches are the steady state the counters are read from (tools/profile_report.py)
    import bench
    native = importlib.import_module("3dgrut_amd.native")
    fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
    sc = getattr(scenes, fn)(**dict(kw, n=n))
    nm = native.NativeGaussianModel(sc, device=dev, spatial_order=True)
    ts = native.NativeTrainStep(nm, tr, scene_extent=extent, overlap_optimizer=True)
    bench.synthetic_optimizer_state(ts)   # the bench's mid-training optimiser state
    ts.tune_placement()                    # ... and its placement tuning
    b.intrinsics_OpenCVPinholeCameraModelParameters = cams.pinhole_intrinsics_dict(W, H, fx, fx)
    ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev, False)
    b.rays_ori, b.rays_dir = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
    b.rgb_gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(1)).to(dev)
    for k in range(iters):
        b.T_to_world = torch.as_tensor(c2ws[k % 8])[None]
        ts.step(b)
for _ in range(0 if step else iters):
    if bwd:
        out = tr.render(model, b, train=True)
        (out["pred_rgb"].mean() + out["pred_opacity"].mean()).backward()
        model.zero_grad(set_to_none=True)
    else:
        with torch.no_grad():
            tr.render(model, b)
torch.cuda.synchronize()
print(tr.tracer_wrapper.stats(), {k: round(v, 3) for k, v in tr.tracer_wrapper.kernel_times().items()})
