"""Dev tool: forward-render variants of the outdoor stand-in scene and print N,V,M,E_f + kernel times."""
import importlib, sys, math, json, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
cams = importlib.import_module("3dgrut_amd.cameras"); model_mod = importlib.import_module("3dgrut_amd.model")
dev = "cuda:0"
W, H, fx = 1237, 822, 1040.0
ro, rd = cams.pinhole_rays(W, H, fx, fx)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
variants = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [dict()]
fn = sys.argv[2] if len(sys.argv) > 2 else "scene_outdoor_like"     # e.g. scene_surface_like
for kw in variants:
    sc = getattr(scenes, fn)(**dict(dict(n=6_000_000, seed=2), **kw))
    model = model_mod.GaussianModel(sc, device=dev)
    tr = gut.Tracer({"render": {"enable_kernel_timings": True}})
    for az in (7.0, 97.0):
        c2w = cams.orbit_c2w(4.5, az, 12.0)
        b = gut.Batch(rays_ori=torch.as_tensor(ro, device=dev), rays_dir=torch.as_tensor(rd, device=dev),
                      T_to_world=torch.as_tensor(c2w, device=dev)[None], intrinsics_OpenCVPinholeCameraModelParameters=K)
        with torch.no_grad():
            for _ in range(3):
                out = tr.render(model, b)
        st = tr.tracer_wrapper.stats(); kt = tr.tracer_wrapper.kernel_times()
        op = out["pred_opacity"].mean().item(); hc = out["hits_count"].mean().item()
        # Gaussians the forward walked (the rows a backward can give a gradient to): distinct ids among the ordered prefixes
        oid = tr.tracer_wrapper.debug_buffer("ordered_ids")
        walked = int(torch.unique(oid[oid != -1]).numel())     # int32 view of the u32 list, padding 0xFFFFFFFF
        print(json.dumps(dict(kw=kw, az=az, V=st["num_visible"], M=st["num_intersections"], Ef=st["traversed_fwd"], walked_gaussians=walked,
                              E_over_M=round(st["traversed_fwd"] / max(1, st["num_intersections"]), 3), walked_over_V=round(walked / max(1, st["num_visible"]), 3),
                              M_over_V=round(st["num_intersections"] / max(1, st["num_visible"]), 2),
                              mean_opacity=round(op, 3), mean_hits=round(hc, 1), ms={k: round(v, 3) for k, v in kt.items() if v >= 0})))
    del model, tr
