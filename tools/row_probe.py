"""Dev probe: the worst per-row gradient deviations of a workload frame against the oracle, with the row's parameters."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_workloads import _frame, oracle, DEV
from tests.common import ROW_REL, ROW_ABS, ROW_NOISE, ROW_FLIP, ROW_FLIP_BOUND
wl = sys.argv[1] if len(sys.argv) > 1 else "garden_like_5M_1297x840"
fr = _frame(wl)
W, H, st, raster = fr["W"], fr["H"], fr["stepper"], fr["tracer"].tracer_wrapper
rgba, dist, hits, vis = st.forward(fr["batch"])
n = fr["model"].num_gaussians
act = st.activate().cpu().numpy(); sph = fr["model"].features.cpu().numpy()
ocam = fr["view"]["oracle_cam"]
ref = oracle.forward(ocam, W, H, act, sph, fr["view"]["ro"], fr["view"]["rd"], sh_degree=3)
rng = np.random.default_rng(11)
rgba_grad = rng.normal(size=(H, W, 4)).astype(np.float32)
dist_grad = (0.05 * rng.normal(size=(H, W, 1))).astype(np.float32)
dens_g, sph_g, _, budget = oracle.backward(ocam, ref, rgba_grad, dist_grad, flip_bound=ROW_FLIP_BOUND)
b, sensor, poses, rgba_, dist_ = st._ctx
g12 = torch.empty((n, 12), dtype=torch.float32, device=DEV); g48 = torch.empty((n, 48), dtype=torch.float32, device=DEV)
def bwd(dg):
    raster.trace_bwd(st.step_id, 3, st.act, fr["model"].features, b.rays_ori.contiguous(), b.rays_dir.contiguous(), None, sensor,
                     poses.timestamps_us[0], poses.timestamps_us[1], poses.T_world_sensors[0], poses.T_world_sensors[1], rgba_,
                     torch.as_tensor(rgba_grad, device=DEV), dist_, dg, out=(g12, g48))
    return g12.cpu().numpy().astype(np.float64)
a = bwd(torch.as_tensor(dist_grad, device=DEV))
a2 = bwd(torch.as_tensor(dist_grad, device=DEV))
print("run-to-run (float atomics) max row diff / scale:", np.linalg.norm((a - a2)[:, :3], axis=1).max())
got, refp = a[:, 0:3], dens_g[:, 0:3]
nr = np.linalg.norm(refp, axis=1); err = np.linalg.norm(got - refp, axis=1)
scale = np.quantile(nr[nr > 0], 0.99)
tight = ROW_REL * nr + ROW_ABS * scale + ROW_NOISE * budget[:, 5]
ratio = err / (tight + ROW_FLIP * budget[:, 0])
cam = np.asarray(fr["view"]["c2w"])[:3, 3]
for i in np.argsort(-ratio)[:8]:
    d = np.linalg.norm(act[i, :3] - cam)
    print(f"row {i}: ratio {ratio[i]:.2f} err {err[i]:.4g} nr {nr[i]:.4g} scale {scale:.4g} flip {budget[i,0]:.3g} noise {budget[i,5]:.3g} "
          f"tiles {ref['tiles_count'][i]} dens {act[i,3]:.4f} scl {act[i,8:11]} dist {d:.2f}\n   ref {refp[i]}\n   gpu {got[i]}\n   dens ref {dens_g[i,3]:.5g} gpu {a[i,3]:.5g}; rot ref {dens_g[i,4:8]} gpu {a[i,4:8]}")
