"""Dev probe (GPU): does the step time depend on WHERE the trainer's state tensors live?  Measures the bench step, then moves
the parameters / moments / activations to freshly allocated memory (old blocks kept alive so they are not reused) and measures
again, several times, inside one process."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
cams = importlib.import_module("3dgrut_amd.cameras"); native = importlib.import_module("3dgrut_amd.native")
dev = "cuda:0"
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS["bicycle_like_6M_1237x822"]
sc = getattr(scenes, fn)(**kw)
nm = native.NativeGaussianModel(sc, device=dev, spatial_order=True)
tr = gut.Tracer({"render": {"enable_kernel_timings": True}})
ts = native.NativeTrainStep(nm, tr, scene_extent=extent, overlap_optimizer=True)
bench.synthetic_optimizer_state(ts)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev, False)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
gt = torch.rand((1, H, W, 3), generator=torch.Generator().manual_seed(1)).to(dev)
ro_t, rd_t = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
def batch(k):
    return gut.Batch(rays_ori=ro_t, rays_dir=rd_t, T_to_world=torch.as_tensor(c2ws[k % 8])[None], rgb_gt=gt, intrinsics_OpenCVPinholeCameraModelParameters=K)
def measure(steps=16, warm=6):
    for k in range(warm): ts.step(batch(k))
    torch.cuda.synchronize(); tr.tracer_wrapper.kernel_times_mean()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(steps): ts.step(batch(k))
    e1.record(); torch.cuda.synchronize()
    kt, _ = tr.tracer_wrapper.kernel_times_mean()
    return e0.elapsed_time(e1) / steps, kt["optimizer_early_2"], kt["optimizer"]
import ctypes as C
lib = ts._lib
flags0 = torch.zeros(((nm.num_gaussians + 63) // 64,), dtype=torch.uint8, device=dev)
zero12 = (C.c_float * 12)(*([0.0] * 12)); zero48 = (C.c_float * 48)(*([0.0] * 48))
def noop_pass_ms(reps=3):
    """The side-stream kernel over EVERY row with zero learning rates and beta = 1, no bias correction: reads and rewrites
    p, m, v and the activations with the values they have (exactly), at the bandwidth this placement allows."""
    st = torch.cuda.current_stream().cuda_stream
    def once():
        rc = lib.gut_adam_unwalked_waves(C.c_void_p(st), nm.num_gaussians, flags0.data_ptr(), nm.raw.data_ptr(), ts.m12.data_ptr(),
                                         ts.v12.data_ptr(), nm.features.data_ptr(), ts.m48.data_ptr(), ts.v48.data_ptr(), zero12, zero48,
                                         1.0, 1.0, ts.eps, 0, ts.act.data_ptr())
        assert rc == 0
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
chk = nm.raw.clone(); chk2 = ts.m48.clone()
print("no-op pass", round(noop_pass_ms(), 3), "unchanged:", bool(torch.equal(chk, nm.raw) and torch.equal(chk2, ts.m48)), flush=True)
keep = []
print("initial", "no-op", round(noop_pass_ms(), 3), [round(x, 3) for x in measure()], flush=True)
for r in range(7):
    for obj, name in ((nm, "raw"), (nm, "features"), (ts, "m12"), (ts, "v12"), (ts, "m48"), (ts, "v48"), (ts, "act")):
        old = getattr(obj, name); keep.append(old); setattr(obj, name, old.clone())
    ts._act_key = None
    print("reallocated", r, "no-op", round(noop_pass_ms(), 3), [round(x, 3) for x in measure()], [hex(getattr(ts, n).data_ptr() >> 21) for n in ("m48", "v48")], flush=True)
