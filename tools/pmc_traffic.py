"""Dev tool: turn the output of tools/collect_profiles.sh (gpurun_out/profiles) into profiles/<round>/ files:
bench_kernel_stats.csv, bench_under_rocprof.json, bench_plain.json, pmc_traffic.json."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "profiles")
dst = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "round1")
os.makedirs(dst, exist_ok=True)
KEYS = {"k_project_on_tiles": "project", "k_expand_tiles": "expand", "k_render(": "render", "k_render_backward": "render_bwd",
        "k_project_backward": "project_bwd", "k_sh_adam": "optimizer"}


def key_of(name):
    for k, v in KEYS.items():
        if k in name:
            return v
    return None


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(dst, "bench_kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "bench_plain.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (tools/fwd_once.py 6000000 2 step: two full native train steps, bicycle-like "
               "stand-in, 1 MI355X), mean per launch. FETCH_SIZE/WRITE_SIZE are in KiB. Per MI355X_MICROARCH.md (HBM section), on "
               "gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.",
       "kernels": {}}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(src, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != ctr:
            continue
        k = key_of(r["Kernel_Name"])
        if k:
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out["kernels"].setdefault(k, {})[ctr + "_KiB"] = sum(v) / len(v)
for k, d in out["kernels"].items():
    d["hbm_bytes"] = 2 * d.get("FETCH_SIZE_KiB", 0.0) * 1024 + d.get("WRITE_SIZE_KiB", 0.0) * 1024
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
