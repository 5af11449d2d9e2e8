"""Dev tool (GPU box): render a bench workload's view 0, compare with the oracle, and dump every pixel whose colour or hit count
differs although the oracle's decision margins call it calm — with the oracle's own walk of that ray in all three fp32 evaluations.
usage: python tools/outlier_probe.py WORKLOAD > gpurun_out/.../outliers.json"""
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tests.common import FLIP_MARGIN_BOUND, COLOUR_TOL
from tests.test_gpu_workloads import _frame
oracle = importlib.import_module("oracle.oracle")
workload = sys.argv[1]
fr = _frame(workload)
W, H, st, raster = fr["W"], fr["H"], fr["stepper"], fr["tracer"].tracer_wrapper
rgba, dist, hits, vis = st.forward(fr["batch"])
act = st.activate().cpu().numpy(); sph = fr["model"].features.cpu().numpy()
ocam = fr["view"]["oracle_cam"]
ref = oracle.forward(ocam, W, H, act, sph, fr["view"]["ro"], fr["view"]["rd"], sh_degree=3)
margins = oracle.render_margins(ocam, ref)
m = margins.min(-1)
g = rgba.cpu().numpy().reshape(H, W, 4); hg = hits.cpu().numpy().reshape(H, W)
diff = np.abs(g - ref["rgba"]).max(-1)
out = ((diff > COLOUR_TOL) | (hg != ref["hits"].reshape(H, W))) & (m >= FLIP_MARGIN_BOUND)
res = []
for py, px in zip(*np.nonzero(out)):
    rec = dict(px=int(px), py=int(py), margin=[float(x) for x in margins[py, px]], diff=float(diff[py, px]), hits_gpu=float(hg[py, px]), hits_ref=float(ref["hits"][py, px, 0]),
               rgba_gpu=[float(x) for x in g[py, px]], rgba_ref=[float(x) for x in ref["rgba"][py, px]])
    walks = {}
    for v in (0, 1, 2):
        with oracle.variant(v):
            walks[v] = oracle.debug_ray(ocam, ref, px, py)
    n = min(len(w) for w in walks.values())
    ent = []
    for k in range(n):
        a = walks[0][k]
        flips = [int(walks[v][k][5]) for v in (0, 1, 2)]
        thr_m = min(abs(a[2] - 0.0113) / (0.0113 * 5.96e-8 * a[4]), abs(a[2] * act[int(a[0]), 3] - 1 / 255) / (1 / 255 * 5.96e-8 * a[4]))
        if thr_m < 200 or len(set(flips)) > 1:
            ent.append(dict(k=k, id=int(a[0]), d2=[float(walks[v][k][1]) for v in (0, 1, 2)], resp=[float(walks[v][k][2]) for v in (0, 1, 2)], alpha=float(a[3]), sigma=float(act[int(a[0]), 3]),
                            nu=float(a[4]), gn=float(a[7]), accepted=flips, margin=float(thr_m), scale=[float(x) for x in act[int(a[0]), 8:11]]))
    rec["near_threshold_entries"] = ent
    rec["walk_lengths"] = [len(walks[v]) for v in (0, 1, 2)]
    res.append(rec)
print(json.dumps(dict(workload=workload, outliers=int(out.sum()), records=res), indent=1))
