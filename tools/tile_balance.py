"""Dev tool: dump per-tile list lengths / traversal depths of the bench view and simulate block scheduling orders."""
import importlib, sys, os, heapq
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gut = importlib.import_module("3dgrut_amd"); scenes = importlib.import_module("3dgrut_amd.scenes")
cams = importlib.import_module("3dgrut_amd.cameras"); model_mod = importlib.import_module("3dgrut_amd.model")
dev = "cuda:0"
W, H, fx = 1237, 822, 1040.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
ro, rd = cams.pinhole_rays(W, H, fx, fx); K = cams.pinhole_intrinsics_dict(W, H, fx, fx)
sc = scenes.scene_outdoor_like(n=n, seed=2)
model = model_mod.GaussianModel(sc, device=dev)
tr = gut.Tracer({"render": {"enable_kernel_timings": True}})
b = gut.Batch(rays_ori=torch.as_tensor(ro, device=dev), rays_dir=torch.as_tensor(rd, device=dev),
              T_to_world=torch.as_tensor(cams.orbit_c2w(4.5, 7.0, 12.0), device=dev)[None], intrinsics_OpenCVPinholeCameraModelParameters=K)
out = tr.render(model, b, train=True)
(out["pred_rgb"].mean() + out["pred_opacity"].mean()).backward()
torch.cuda.synchronize()
r = tr.tracer_wrapper
rng = r.debug_buffer("tile_ranges").cpu().numpy().view(np.uint32).reshape(-1, 2)
tf = r.debug_buffer("tile_traversed_fwd").cpu().numpy().view(np.uint32)
tb = r.debug_buffer("tile_traversed_bwd").cpu().numpy().view(np.uint32)
ln = (rng[:, 1] - rng[:, 0]).astype(np.int64)
np.savez(os.path.join("gpurun_out", "tile_balance.npz"), length=ln, trav_fwd=tf, trav_bwd=tb)
print("tiles", len(ln), "len mean/max", ln.mean(), ln.max(), "trav_fwd mean/max", tf.mean(), tf.max(), "trav_bwd mean/max", tb.mean(), tb.max())


def simulate(work, order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for t in order:
        s = heapq.heappop(h)
        heapq.heappush(h, s + work[t] + 30.0)   # +fixed per-block overhead in "entries"
    return max(h)


for name, w in (("fwd", tf.astype(np.float64)), ("bwd", tb.astype(np.float64))):
    for slots in (256 * 2, 256 * 4, 256 * 8):
        ideal = (w.sum() + 30.0 * len(w)) / slots
        nat = simulate(w, range(len(w)), slots)
        lpt = simulate(w, np.argsort(-w), slots)
        lpt_len = simulate(w, np.argsort(-ln), slots)
        print(f"{name} slots {slots}: ideal {ideal:.0f} natural {nat:.0f} lpt(true) {lpt:.0f} lpt(list length) {lpt_len:.0f} max tile {w.max():.0f}")
print(r.kernel_times())
