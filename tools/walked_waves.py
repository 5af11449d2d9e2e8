"""Dev probe (GPU): on the bench frame, which 64-row waves of the Morton-ordered model contain a Gaussian the forward WALKED
(list entries before whole-tile termination)?  Only those can receive a gradient from the backward."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
cams = importlib.import_module("3dgrut_amd.cameras"); native = importlib.import_module("3dgrut_amd.native")
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "bicycle_like_6M_1237x822"
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS[name]
sc = getattr(scenes, fn)(**kw)
nm = native.NativeGaussianModel(sc, device=dev, spatial_order=True)
tr = gut.Tracer({"render": {}})
ts = native.NativeTrainStep(nm, tr, scene_extent=extent, overlap_optimizer=False)
ro, rd, c2ws = bench.make_views(cams, 8, W, H, fx, radius, elev, False)
K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
for v in range(2):
    b = gut.Batch(rays_ori=torch.as_tensor(ro, device=dev), rays_dir=torch.as_tensor(rd, device=dev), T_to_world=torch.as_tensor(c2ws[v])[None],
                  intrinsics_OpenCVPinholeCameraModelParameters=K)
    ts.forward(b)
    r = ts.raster
    ranges = r.debug_buffer("tile_ranges").view(-1, 2).long()
    trav = r.debug_buffer("tile_traversed_fwd").long()
    ids = r.debug_buffer("ordered_ids").long()
    cnt = r.debug_buffer("tiles_count")
    n = cnt.numel()
    start = ranges[:, 0]
    total = int(trav.sum())
    tile_of = torch.repeat_interleave(torch.arange(trav.numel(), device=dev), trav)
    off = torch.arange(total, device=dev) - torch.repeat_interleave(torch.cumsum(trav, 0) - trav, trav)
    walked_ids = ids[start[tile_of] + off]
    walked_ids = walked_ids[walked_ids < n]
    walked = torch.zeros(n, dtype=torch.bool, device=dev); walked[walked_ids] = True
    has = cnt != 0
    pad = (-n) % 64
    w_has = torch.nn.functional.pad(has, (0, pad)).view(-1, 64).any(1)
    w_walk = torch.nn.functional.pad(walked, (0, pad)).view(-1, 64).any(1)
    print(f"view {v}: N {n}, rows with tiles {int(has.sum())}, walked entries {total}, walked unique rows {int(walked.sum())}")
    print(f"  waves: total {w_has.numel()}, without tiles {int((~w_has).sum())}, with tiles but no walked row {int((w_has & ~w_walk).sum())}, with a walked row {int(w_walk.sum())}")
    for g in (256,):
        padg = (-n) % g
        gw = torch.nn.functional.pad(walked, (0, padg)).view(-1, g).any(1)
        print(f"  {g}-row blocks with a walked row: {int(gw.sum())} of {gw.numel()}")
