"""Dev probe (GPU): where the drop-in train step (Tracer.render -> _Autograd -> torch.optim.Adam) spends its time."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes"); cams = importlib.import_module("3dgrut_amd.cameras")
model_mod = importlib.import_module("3dgrut_amd.model"); train_mod = importlib.import_module("3dgrut_amd.train")
dev = torch.device("cuda", 0)
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
scene = getattr(scenes, fn)(**kw)
tracer = gut.Tracer({"render": {"enable_kernel_timings": True}})
model = model_mod.GaussianModel(scene, device=dev, sh_degree=3)
stepper = train_mod.TrainStep(model, tracer, scene_extent=extent)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev)
ro_t, rd_t = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
gt = torch.rand((1, H, W, 3), device=dev)
def batch(i):
    return gut.Batch(rays_ori=ro_t, rays_dir=rd_t, T_to_world=torch.as_tensor(c2ws[i % 8])[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K)
for i in range(3): stepper.step(batch(i))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8): stepper.step(batch(i))
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/step")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(4): stepper.step(batch(i))
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
