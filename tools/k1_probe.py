import importlib, sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
cams = importlib.import_module("3dgrut_amd.cameras"); model_mod = importlib.import_module("3dgrut_amd.model")
dev = "cuda:0"
W, H, fx = 1237, 822, 1040.0
ro, rd = cams.pinhole_rays(W, H, fx, fx); K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
sc = scenes.scene_outdoor_like(n=6_000_000, seed=2)
c2w = cams.orbit_c2w(4.5, 7.0, 12.0)
def run(tag, sc, conf):
    model = model_mod.GaussianModel(sc, device=dev)
    tr = gut.Tracer(conf)
    b = gut.Batch(rays_ori=torch.as_tensor(ro, device=dev), rays_dir=torch.as_tensor(rd, device=dev),
                  T_to_world=torch.as_tensor(c2w, device=dev)[None], intrinsics_OpenCVPinholeCameraModelParameters=K)
    with torch.no_grad():
        for _ in range(4): tr.render(model, b)
    st = tr.tracer_wrapper.stats(); kt = tr.tracer_wrapper.kernel_times()
    print(tag, dict(V=st["num_visible"], M=st["num_intersections"]), {k: round(v, 3) for k, v in kt.items() if v >= 0}, flush=True)
run("default", sc, {"render": {"enable_kernel_timings": True}})
run("no_tile_culling", sc, {"render": {"enable_kernel_timings": True, "splat": {"tile_based_culling": False}}})
sc2 = {k: v.copy() for k, v in sc.items()}; sc2["positions"][:, :] += 1000.0   # everything culled by z / out of view
run("all_culled", sc2, {"render": {"enable_kernel_timings": True}})
sc3 = {k: v.copy() for k, v in sc.items()}; sc3["density"][:] = 0.001          # culled by opacity before any math
run("opacity_culled", sc3, {"render": {"enable_kernel_timings": True}})
