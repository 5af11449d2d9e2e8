#!/usr/bin/env python
"""bench.py — 3DGUT train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of the hot path on one view per GPU: HIP forward render,
0.8*L1+0.2*(1-SSIM) loss, HIP backward, [RCCL all-reduce of the Gaussian gradients for N>1], Adam
update of all 59 parameters per Gaussian (trainer.py:705-778 call sequence).  Workload (N=1): the
configuration BASELINE.json's metric is quoted on — "MipNeRF360 bicycle, 1237x822, ~6M Gaussians" —
as a seeded synthetic stand-in (no datasets/checkpoints in the environment): scene_outdoor_like(6e6).
Weak scaling: every rank renders its own view of the replicated scene each step.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  "roofline":     dominant HBM-bound kernel's algorithmic bytes / its mean hipEvent duration over the timed region,
  "roofline_valu": the same for the step's longest launch when that is one of the two compositors, which are bound by
                   vector-instruction issue (wave64 instructions per second against the chip's issue peak),
  "cpu_baseline": pure-PyTorch per-ray composite (oracle/per_ray_torch.py) on a bounded ray sample,
plus "render_ms_per_frame" (forward-only, event time around the whole forward incl. sort + count readback,
the reference's `forward_render` definition) and the scene statistics N,V,M,T,P,E_f,E_b.
"""
import argparse
import gc
import importlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene fn name, kwargs, W, H, fx, camera radius, elevation, scene extent)
    "bicycle_like_6M_1237x822": ("scene_outdoor_like", dict(n=6_000_000, seed=2), 1237, 822, 1040.0, 4.5, 12.0, 5.0),
    "lego_like_300k_800x800": ("scene_lego_like", dict(n=300_000, seed=1), 800, 800, 1111.1, 4.0, 25.0, 1.3),
    "c1_1k_128x128": ("scene_c1", dict(n=1000, seed=0), 128, 128, 128.0, 4.0, 0.0, 1.0),
    # BASELINE configs[3]: ScanNet++-DSLR-like OpenCV fisheye (zero radial coefficients, as its loader sets them), camera
    # inside the scene; fx is the fisheye focal length (equidistant model): ~150 degrees across the 1752-pixel width
    "scannetpp_like_fisheye_300k_1752x1168": ("scene_lego_like", dict(n=300_000, seed=3), 1752, 1168, 660.0, 0.6, 10.0, 1.3),
    # workload sensitivity (not a BASELINE config): the bicycle stand-in with SURVEY.md §8d C3's literal parameters
    "bicycle_like_6M_survey_c3": ("scene_outdoor_like", dict(n=6_000_000, seed=2, scale_mu=math.log(0.01), opacity_logit_mean=0.0,
                                                           opacity_logit_std=1.5), 1237, 822, 1040.0, 4.5, 12.0, 5.0),
    # the OTHER end of the headline's range: the same size and cameras, every Gaussian on a 2-D shell as a flat,
    # mostly faint disc: E/M = 0.4 - 0.5 instead of 0.12, four times the compositing work per frame (scenes.scene_surface_like)
    "bicycle_like_6M_surface": ("scene_surface_like", dict(n=6_000_000, seed=6), 1237, 822, 1040.0, 4.5, 12.0, 5.0),
    # BASELINE configs[4]: MipNeRF360-garden-like, one view per GPU
    "garden_like_5M_1297x840": ("scene_outdoor_like", dict(n=5_000_000, seed=4), 1297, 840, 1090.0, 4.2, 15.0, 5.0),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_GINSTR_S = 1024 * 2.4 / 2.0   # wave64 VALU instructions per ns, chip-wide: one per SIMD every 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
VALU_ISSUE_NS = 1.16   # one wave64 VALU instruction per SIMD every 1.16 ns with >= 2 waves resident (tools/pk_rate.hip, measured)


def algorithmic_bytes(st, kernel, end_bit):
    """SURVEY.md §8d byte model, per kernel launch.  st["lazy_moments"]: the zero-gradient update of a wave that cannot receive a
    gradient reads p, m, v (720 B per row) and writes p and the next activation only (240 + 48 B): 1008 instead of 1488 B."""
    Z = 1008 if st.get("lazy_moments") else 1488
    N, V, M, T, P = (st[k] for k in ("num_particles", "num_visible", "num_intersections", "num_tiles", "num_pixels"))
    Ef, Eb = st["traversed_fwd"], st["traversed_bwd"]
    b = math.ceil((end_bit - 32) / 8)  # lazy tile order (default): onesweep passes over the tile bits only (csrc/gut_sort.hip)
    return {
        "project": (48 + 44) * N + 204 * V,
        "scan": 8 * N,
        "expand": 8 * N + 36 * V + 12 * M,
        "sort": (8 + 24 * b) * M,
        "ranges": 8 * M + 8 * T,
        "render": 8 * T + 64 * Ef + 48 * P,
        "render_bwd": 8 * T + 64 * Eb + 64 * P + 112 * V,
        "project_bwd": (40 + 384) * V + 252 * N,
        # one-pass optimiser (k_sh_adam<scratch>): raw p/m/v in+out 288, next activation 48, SH p/m/v in+out 1152, count 4 per
        # Gaussian; 64-byte gradient row read, 12-byte RGB, and the 64-byte row zeroed again (replaces the per-step clear of the
        # whole gradient buffer) per Gaussian with tiles
        # (one-pass form with lazy moments: the rows of gradient-free waves, counted by the library as side_stream_rows when the
        #  side stream ran, move Z instead of 1488 bytes; without that count the figure is the eager upper bound)
        "optimizer": 1492 * N + (76 + 64) * V,
        # split form (default at one view): R = st["side_stream_rows"] rows (whole 64-row waves: those without any tile, and part
        # of those the forward walked no Gaussian of; counted by the library for the last step) are updated by
        # k_adam_rows_without_gradient on a side stream under the compositing kernels, k_sh_adam then walks the other N - R rows
        # (at most V of them have a gradient row); both read every tile count (4 N)
        "optimizer_late": 4 * N + 1488 * (N - (st.get("side_stream_rows") or (N - V))) + (76 + 64) * min(V, N - (st.get("side_stream_rows") or (N - V))),
        "optimizer_early": 4 * N + Z * (st.get("side_stream_rows") or (N - V)),
        # its second launch alone: walks every row block once the forward's walked set is known (tile counts + one byte per wave)
        "optimizer_early_2": 4 * N + N // 64 + Z * max(0, (st.get("side_stream_rows") or 0) - (st.get("side_stream_rows_first_launch") or 0)),
    }[kernel]


def make_views(cams, n_views, W, H, fx, radius, elev, fisheye=False):
    ro, rd = cams.fisheye_rays(W, H, fx, fx) if fisheye else cams.pinhole_rays(W, H, fx, fx)
    views = []
    for i in range(n_views):
        c2w = cams.orbit_c2w(radius, 360.0 * i / n_views + 7.0, elev)
        views.append(c2w)
    return ro, rd, views


def cpu_baseline(scene, cams_mod, pose_mod, W, H, fx, c2w, sh_degree, budget_s=30.0, fisheye=False):
    """The CPU restatement of the path (oracle/gut_oracle.c, gcc -O2 -fopenmp) on ONE full frame of the same workload:
    projection + binning + sort + compositing forward, then the compositing / projection backward.  No loss, no optimiser:
    it is the renderer's share of a train step, so the ratio to `value` flatters the CPU.  If the forward alone exceeds the
    budget the backward is not run and the forward time is doubled as an estimate (said so in `sample`)."""
    oracle = importlib.import_module("oracle.oracle")
    scn = importlib.import_module("3dgrut_amd.scenes")
    tq = pose_mod.sensor_pose_from_c2w(c2w).T_world_sensors[0]
    if fisheye:
        K = cams_mod.fisheye_intrinsics_dict(W, H, fx, fx)
        ocam = dict(model="fisheye", principal_point=K["principal_point"], focal_length=K["focal_length"],
                    radial=list(K["radial_coeffs"]), max_angle=K["max_angle"], pose_start=tq)
        ro, rd = cams_mod.fisheye_rays(W, H, fx, fx)
    else:
        K = cams_mod.pinhole_intrinsics_dict(W, H, fx, fx)
        ocam = dict(model="pinhole", principal_point=K["principal_point"], focal_length=K["focal_length"], radial=K["radial_coeffs"],
                    tangential=K["tangential_coeffs"], thin_prism=K["thin_prism_coeffs"], pose_start=tq)
        ro, rd = cams_mod.pinhole_rays(W, H, fx, fx)
    d12 = scn.pack_density(scene)
    threads = int(oracle.lib().oracle_max_threads())
    t0 = time.time()
    fwd = oracle.forward(ocam, W, H, d12, scene["features"], ro, rd, sh_degree=sh_degree)
    tf = time.time() - t0
    if tf <= budget_s / 2:
        g = np.random.default_rng(0).normal(size=(H, W, 4)).astype(np.float32)
        t0 = time.time()
        oracle.backward(ocam, fwd, g, np.zeros((H, W, 1), np.float32))
        tb = time.time() - t0
        how = f"forward {tf:.2f} s + backward {tb:.2f} s"
    else:
        tb = tf
        how = f"forward {tf:.2f} s; backward not run (budget), counted as another {tf:.2f} s"
    return {"value": 1.0 / (tf + tb), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"one full {W}x{H} frame of the {d12.shape[0]}-Gaussian workload, render forward+backward only (no loss, no "
                      f"optimiser), C restatement oracle/gut_oracle.c with OpenMP on {threads} threads: {how}; "
                      f"M = {fwd['M']} intersections"}


def cpu_baseline_per_ray_torch(scene, cams_mod, pose_mod, W, H, fx, c2w, sh_degree, budget_s=8.0):
    """Pure-PyTorch per-ray composite (oracle/per_ray_torch.py), forward + autograd backward, fp32, on the rays of a
    centred crop; the Gaussian set is culled with the UT projection rule to the ones whose 2-D extent touches the
    crop (otherwise brute force over all 6 M).  The crop doubles until ~budget_s of CPU work is reached."""
    prt = importlib.import_module("oracle.per_ray_torch")
    # 32 threads, not every core: the composite is a chain of small per-chunk tensor ops (1024 rays x 1024 Gaussians), and torch's
    # CPU kernels get SLOWER beyond a few dozen threads on them — measured on a 256-thread GPU box: 13 s for a 4x4 crop with 256
    # threads (1.2e-6 images/s) against 14 s for a 128x128 crop with 32 (1.1e-3 images/s)
    threads = max(1, min(os.cpu_count() or 1, 32))
    torch.set_num_threads(threads)
    tq = pose_mod.sensor_pose_from_c2w(c2w).T_world_sensors[0]
    cam = dict(model="pinhole", principal_point=[W / 2, H / 2], focal_length=[fx, fx])
    ro, rd = cams_mod.pinhole_rays(W, H, fx, fx)
    params = {k: torch.tensor(v) for k, v in scene.items()}
    pr = prt.project(cam, tq, W, H, params, dtype=torch.float32)
    crop, best = 4, None
    while True:
        x0, y0 = W // 2 - crop // 2, H // 2 - crop // 2
        c, e = pr["center"], pr["extent"]
        inb = pr["valid"] & (c[:, 0] + e[:, 0] >= x0) & (c[:, 0] - e[:, 0] <= x0 + crop) & \
            (c[:, 1] + e[:, 1] >= y0) & (c[:, 1] - e[:, 1] <= y0 + crop)
        idx = torch.nonzero(inb).squeeze(1)
        sub = {k: v[idx].clone().requires_grad_(True) for k, v in params.items()}
        ys, xs = torch.meshgrid(torch.arange(y0, y0 + crop), torch.arange(x0, x0 + crop), indexing="ij")
        pix = (ys * W + xs).reshape(-1)
        t1 = time.time()
        rgba, dist, hits = prt.render_per_ray(cam, tq, W, H, sub, ro, rd, sh_degree=sh_degree, dtype=torch.float32,
                                              pixel_subset=pix, pix_chunk=1024, gauss_chunk=1024)
        rgba.sum().backward()
        dt = time.time() - t1
        best = (crop, dt, int(idx.numel()))
        if dt * 3.5 > budget_s or crop * 2 > min(W, H):
            break
        crop *= 2
    crop, dt, ng = best
    rays = crop * crop
    img_s = 1.0 / (dt * (W * H) / rays)
    return {"value": img_s, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"render fwd+bwd only (no loss/Adam) on a {crop}x{crop} centre crop = {rays} of {W * H} rays vs the "
                      f"{ng} UT-culled Gaussians touching it; pure-PyTorch fp32 per-ray composite + autograd took {dt:.2f} s; "
                      f"value is extrapolated to the full image"}


def launcher_command(argv, n_gpus, port, python=None):
    """The command `bench.py --gpus N` re-issues itself as when it was started without a rank environment: one process per
    GPU over RCCL, exactly the driver's own form (task contract)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(argv, n_gpus):
    """Started as plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE): spawn the N rank processes as CHILDREN of this
    process — which has not touched the GPU and never execs — relay rank 0's JSON line, and exit non-zero if any rank fails."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = launcher_command(argv, n_gpus, _free_port())
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or not json_lines:
        print(f"[bench] multi-GPU run failed (exit code {proc.returncode}, {len(json_lines)} JSON lines)", file=sys.stderr)
        return proc.returncode or 1
    print(json_lines[-1], flush=True)
    return 0


def profile_counters():
    """Per-kernel PMC counters of the tracked profile of this command (profiles/<latest round>/pmc_traffic.json, pmc_sq.json)."""
    out = {}
    for rnd in ("round4", "round3", "round2", "round1"):
        d = os.path.join(ROOT, "profiles", rnd)
        if not os.path.isfile(os.path.join(d, "pmc_traffic.json")):
            continue
        try:
            for k, v in json.load(open(os.path.join(d, "pmc_traffic.json")))["kernels"].items():
                out.setdefault(k, {}).update(v)
            if os.path.isfile(os.path.join(d, "pmc_sq.json")):
                for k, v in json.load(open(os.path.join(d, "pmc_sq.json")))["kernels"].items():
                    out.setdefault(k, {}).update(v)
            out["_source"] = f"profiles/{rnd}"
            return out
        except Exception:
            out = {}
    return out


def data_label(args):
    """The `data` field of the JSON line."""
    if getattr(args, "ply", None):
        return "ply:" + os.path.basename(args.ply) + ((" colmap:" + os.path.basename(os.path.normpath(args.colmap))) if args.colmap else "")
    return "synthetic"


def load_scene(args, workload, kw=None):
    """(scene dict, ColmapScene or None, scene extent) of a run: the named workload's seeded stand-in, or — `--ply PATH` — a real
    scene in its place ("real scenes substitute 1:1 if supplied as PLY"; threedgrut/model/model.py:671-719 layout),
    optionally with a COLMAP directory's training views instead of the workload's orbit cameras (`--colmap DIR`)."""
    scenes = importlib.import_module("3dgrut_amd.scenes")
    fn, wkw, W, H, fx, radius, elev, extent = WORKLOADS[workload]
    if getattr(args, "ply", None) and workload == args.workload:
        io_ply = importlib.import_module("3dgrut_amd.io_ply")
        scene = io_ply.scene_from_ply(args.ply)
        colmap = None
        if getattr(args, "colmap", None):
            io_colmap = importlib.import_module("3dgrut_amd.io_colmap")
            colmap = io_colmap.ColmapScene(args.colmap, split="train", downsample_factor=getattr(args, "colmap_downsample", 1))
            if len(colmap) == 0:
                raise SystemExit(f"--colmap {args.colmap}: no training views")
            extent = colmap.cameras_extent
        return scene, colmap, extent
    return getattr(scenes, fn)(**(wkw if kw is None else kw)), None, extent  # same seed on every rank -> identical replicas


def synthetic_optimizer_state(stepper):
    """Non-zero Adam moments on every row (see run_workload)."""
    g_state = torch.Generator(device=stepper.m12.device).manual_seed(7)
    for m_, v_ in ((stepper.m12, stepper.v12), (stepper.m48, stepper.v48)):
        m_.normal_(0.0, 1e-6, generator=g_state)
        v_.fill_(1e-8)


def run_workload(args, env, workload, steps, warmup, render_frames):
    """Build the workload, run `warmup` untimed and `steps` timed train steps (barrier + synchronize on both sides, max over
    ranks), then `render_frames` forward-only frames.  Returns the measurements and the objects the report needs."""
    rank, world, dev, dist = env["rank"], env["world"], env["dev"], env["dist"]
    gut = importlib.import_module("3dgrut_amd")
    scenes = importlib.import_module("3dgrut_amd.scenes")
    cams = importlib.import_module("3dgrut_amd.cameras")
    pose_mod = importlib.import_module("3dgrut_amd.pose")
    model_mod = importlib.import_module("3dgrut_amd.model")
    train_mod = importlib.import_module("3dgrut_amd.train")
    dp_mod = importlib.import_module("3dgrut_amd.dp")

    fn, kw, W, H, fx, radius, elev, extent = WORKLOADS[workload]
    kw = dict(kw)
    if args.num_gaussians:
        kw["n"] = args.num_gaussians
    scene, colmap, extent = load_scene(args, workload, kw)
    sh_degree = 3
    # (BENCH_NO_TIMING_EVENTS=1: developer A/B of what the per-kernel / per-phase events inside the timed region cost; the line then
    #  carries no per-kernel times and no live roofline)
    no_events = os.environ.get("BENCH_NO_TIMING_EVENTS") is not None
    tracer = gut.Tracer({"render": {"enable_kernel_timings": not no_events}})
    if args.full_sort:
        tracer.tracer_wrapper.set_lazy_tile_order(False)
    if getattr(args, "early_extra", None) is not None:
        tracer.tracer_wrapper.set_early_extra_percent(args.early_extra)
    if args.trainer == "native":
        native_mod = importlib.import_module("3dgrut_amd.native")
        model = native_mod.NativeGaussianModel(scene, device=dev, sh_degree=sh_degree, spatial_order=not args.scene_order)
        stepper = native_mod.NativeTrainStep(model, tracer, scene_extent=extent, world_size=world, selective=args.selective_adam,
                                             rank=rank, fused_sh_adam=not args.dense_exchange,
                                             overlap_optimizer=False if args.no_overlap_optimizer else (True if getattr(args, "force_overlap_optimizer", False) else None),
                                             dp_exchange=args.dp_exchange, dp_side_stream=not args.no_dp_side_stream,
                                             lazy_moments=not getattr(args, "no_lazy_moments", False))
        if not getattr(args, "fresh_optimizer_state", False):
            # Synthetic MID-TRAINING optimiser state, like the synthetic parameters: every row has non-zero Adam moments, as every
            # Gaussian of a trained 6 M scene has (a moment only returns to exact zero after thousands of gradient-free steps),
            # so that nothing keyed on zero moments can flatter the line (82 % of this synthetic scene's rows are never touched
            # by any of the eight views).  |m| / sqrt(v) = 0.01: the parameters drift by less than 0.1 learning-rate steps in
            # total, so the scene statistics stay what they are.
            synthetic_optimizer_state(stepper)
        # (NativeTrainStep.tune_placement runs after the first two warm-up steps, below)
        if getattr(args, "densification_statistics", False):
            # the first half of a reference run (strategy/gs.py:106-115, every iteration until densify.end_iteration = 15000):
            # per-view position-gradient statistics between backward and optimiser
            strategy_mod = importlib.import_module("3dgrut_amd.strategy")
            res_strategy = strategy_mod.GSStrategy(stepper).attach()
    else:
        model = model_mod.GaussianModel(scene, device=dev, sh_degree=sh_degree)
        stepper = train_mod.TrainStep(model, tracer, scene_extent=extent, world_size=world)

    n_views = max(8, world)
    fisheye = "fisheye" in workload
    ro, rd, c2ws = make_views(cams, n_views, W, H, fx, radius, elev, fisheye)
    ro_t, rd_t = torch.as_tensor(ro, device=dev), torch.as_tensor(rd, device=dev)
    K = cams.fisheye_intrinsics_dict(W, H, fx, fx) if fisheye else cams.pinhole_intrinsics_dict(W, H, fx, fx)
    kkey = "intrinsics_OpenCVFisheyeCameraModelParameters" if fisheye else "intrinsics_OpenCVPinholeCameraModelParameters"
    g = torch.Generator(device="cpu").manual_seed(100)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    gt = torch.stack([0.5 + 0.4 * torch.sin(6.0 * xx), 0.5 + 0.4 * torch.cos(5.0 * yy), 0.5 * (xx + yy)], -1)
    gt = (gt + 0.02 * torch.randn(gt.shape, generator=g)).clamp(0, 1)[None].to(dev)

    colmap_batches = {}

    def batch_for(step):
        v = dp_mod.view_index(step, rank, world, n_views)
        if colmap is not None:   # the scene's own training views (cached on the device); targets: the image if it is there
            i = v % len(colmap)
            if i not in colmap_batches:
                b = colmap.batch(i, device=dev, pose_on_host=args.host_pose)
                if b.rgb_gt is None:
                    hh, ww = b.rays_dir.shape[1], b.rays_dir.shape[2]
                    b.rgb_gt = torch.nn.functional.interpolate(gt.permute(0, 3, 1, 2), size=(hh, ww), mode="bilinear").permute(0, 2, 3, 1).contiguous()
                colmap_batches[i] = b
            return colmap_batches[i]
        # the 4x4 pose stays on the host (the tracer needs it there to fill the camera struct; a device tensor would
        # cost a blocking read-back per step, as in the reference's tracer.py:353-356)
        pose = torch.as_tensor(c2ws[v])[None] if args.host_pose else torch.as_tensor(c2ws[v], device=dev)[None]
        return gut.Batch(rays_ori=ro_t, rays_dir=rd_t, T_to_world=pose, rgb_gt=gt, **{kkey: K})

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    tune = (args.trainer == "native" and stepper.model.num_gaussians >= 1_000_000 and not getattr(args, "no_placement_tuning", False))
    for s in range(warmup):
        stepper.step(batch_for(s))
        if tune and s == min(1, warmup - 1):
            # Re-place the three [N,48] state tensors where that makes the optimiser's stream faster — after the first steps, when
            # the library has allocated its scratch.  (With the tuning's churn of gigabyte-sized allocations BEFORE the first step
            # the compositing kernels ran 4 - 6 % slower in half of the processes, interleaved runs on one box; no single buffer's
            # placement explains it — DESIGN.md §5, "the compositors' speed wanders" — but this order has not shown it.)
            stepper.tune_placement(attempts=getattr(args, "placement_attempts", 4))
            tune = False
    # the trainer's overlap probe times steps 2..9 in alternating forms and decides at step 10: never inside the timed region
    extra_warmup = 0
    while getattr(stepper, "probe_pending", False) and extra_warmup < 12:
        stepper.step(batch_for(warmup + extra_warmup))
        extra_warmup += 1
    warmup += extra_warmup
    raster = tracer.tracer_wrapper
    barrier()
    if not no_events:
        raster.kernel_times_mean()  # reset the per-kernel event ring
    raster.collect_times()
    # The timed region carries NO events: it is the product's default configuration (render.enable_kernel_timings = false).  The two dozen
    # event records per step (every kernel boundary, the phases) cost 5 - 7 % of a 2.4 ms step — measured, BENCH_NO_TIMING_EVENTS and
    # DESIGN.md section 6.  Every per-kernel duration of the line — the roofline's kernel included — is measured right after the timed
    # region by HIP events over a second pass of the same number of steps on the same views (`instrumented_pass`).
    bare_steps = hasattr(raster, "set_kernel_timing_set") and not no_events
    if bare_steps:
        raster.set_kernel_timing_set(2)
    if hasattr(stepper, "phase_timing"):
        stepper.phase_timing = not no_events and not bare_steps
    # A full (generation-2) collection of this process's Python heap is a 40 - 60 ms host pause (seen at a fixed step of a run:
    # 2.6 -> 4.3 ms/step over 30 steps); real training pays it once in thousands of steps.  gc.freeze() moves what exists now out of
    # the collector's sight, so a collection inside the timed region only looks at the steps' own few objects.  (NOT gc.collect():
    # with it K6 / K7 ran 5 % slower in the loop that followed — 2.55 -> 2.67 ms/step, interleaved runs on one box; DESIGN.md §5,
    # "the compositors' speed wanders".)
    gc.freeze()
    t0 = time.perf_counter()
    for s in range(steps):
        stepper.step(batch_for(warmup + s))
    barrier()
    gc.unfreeze()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if no_events:   # developer A/B only: nothing of the line's per-kernel content can be filled in
        print(json.dumps({"metric": "A/B only: step without timing events", "value": world * steps / elapsed,
                          "ms_per_step": 1000.0 * elapsed / steps, "render_ms_per_frame": float("nan"), "per_kernel": {}, "phase_ms": {}}))
        sys.exit(0)
    instrumented_ms = None
    if bare_steps:
        raster.set_kernel_timing_set(0)
        if hasattr(stepper, "phase_timing"):
            stepper.phase_timing = True
        raster.collect_times()
        t1 = time.perf_counter()
        for s in range(steps):
            stepper.step(batch_for(warmup + steps + s))
        barrier()
        instrumented_ms = 1000.0 * (time.perf_counter() - t1) / steps
        phases = stepper.phase_times_mean() if hasattr(stepper, "phase_times_mean") else {}
        if hasattr(stepper, "phase_timing"):
            stepper.phase_timing = False
        ktimes, kcount = raster.kernel_times_mean()
    else:
        phases = stepper.phase_times_mean() if hasattr(stepper, "phase_times_mean") else {}
        if hasattr(stepper, "phase_timing"):
            stepper.phase_timing = False
        ktimes, kcount = raster.kernel_times_mean()
    fb = raster.collect_times()
    stats = raster.stats()
    stats["lazy_moments"] = bool(getattr(stepper, "lazy_moments", False))

    # forward-only render time (reference's FPS definition: mean forward_render ms, threedgrut/render.py:231-251)
    with torch.no_grad():
        for s in range(render_frames):
            stepper.render(batch_for(s), train=False)
    torch.cuda.synchronize(dev)
    render_ms = raster.collect_times().get("forward_render", float("nan"))
    return dict(value=world * steps / elapsed, ms_per_step=1000.0 * elapsed / steps, phases=phases,
                step_spans=getattr(stepper, "last_step_spans_ms", None), ktimes=ktimes, kcount=kcount, fb=fb, instrumented_ms=instrumented_ms,
                stats=stats, render_ms=render_ms, scene=scene, cams=cams, pose_mod=pose_mod, c2ws=c2ws, W=W, H=H, fx=fx,
                fisheye=fisheye, stepper=stepper, extra_warmup=extra_warmup, colmap=colmap is not None, extent=extent,
                batch_for=batch_for, tracer=tracer, dev=dev)


def run_drop_in(res, steps=16, warmup=5, optimizer_type="adam", raw_parameters=True):
    """The SAME workload through the reference's own surface, untouched trainer side: Tracer.render -> Tracer._Autograd ->
    torch loss -> loss.backward() -> torch.optim.Adam (3dgrut_amd/train.TrainStep = trainer.py:705-778 with only the renderer
    swapped).  This is what a user of threedgrut.trainer gets by switching the plugin; `value` above additionally needs the
    trainer to call the gut_optimize_* entry points.  optimizer_type "selective_adam": the reference's other
    optimiser option (configs/base_gs.yaml:82, model.py:512) with ITS plugin swapped too (optimizers.SelectiveAdam)."""
    model_mod = importlib.import_module("3dgrut_amd.model")
    train_mod = importlib.import_module("3dgrut_amd.train")
    dev = res["dev"]
    model = model_mod.GaussianModel(res["scene"], device=dev, sh_degree=3)
    # Tracer.raw_parameters (default True): a model with the reference's activation callables hands its nn.Parameters over as they
    # are and the library applies normalize / exp / sigmoid in-kernel (gut_trace_raw_model_fields); False = the reference's own
    # calls (model.get_rotation() ... as torch kernels, tracer.py:323-327)
    res["tracer"].raw_parameters = bool(raw_parameters)
    stepper = train_mod.TrainStep(model, res["tracer"], scene_extent=res["extent"], world_size=1, optimizer_type=optimizer_type)
    for s in range(warmup):
        stepper.step(res["batch_for"](s))
    torch.cuda.synchronize(dev)
    res["tracer"].tracer_wrapper.collect_times()
    stepper.phase_timing = True
    gc.freeze()
    t0 = time.perf_counter()
    for s in range(steps):
        stepper.step(res["batch_for"](warmup + s))
    torch.cuda.synchronize(dev)
    gc.unfreeze()
    dt = time.perf_counter() - t0
    fb = res["tracer"].tracer_wrapper.collect_times()
    phases = stepper.phase_times_mean()
    # where a drop-in step goes: the trainer's four calls, and inside the first and third the plugin's own share
    phases = {"render_call": phases.get("render"), "of_which_plugin_forward": fb.get("forward_render"), "loss": phases.get("loss"),
              "backward_call": phases.get("backward"), "of_which_plugin_backward": fb.get("backward_render"), "optimizer_step": phases.get("optimizer")}
    form = "the model's pre-activation tensors, activated in-kernel (Tracer.raw_parameters)" if raw_parameters else "torch activations (model.get_rotation() ...), as the reference calls them"
    if optimizer_type == "selective_adam":
        return {"trainer": "autograd (Tracer.render -> _Autograd -> SelectiveAdam.step(mog_visibility)): optimizer.type=selective_adam, both plugins swapped",
                "parameters": form, "value": steps / dt, "unit": "images/s", "ms_per_step": 1000.0 * dt / steps, "steps": steps, "warmup": warmup,
                "phase_ms": phases}
    return {"trainer": "autograd (Tracer.render -> _Autograd -> torch.optim.Adam(fused)), the reference's surface unchanged",
            "parameters": form, "value": steps / dt, "unit": "images/s", "ms_per_step": 1000.0 * dt / steps, "steps": steps, "warmup": warmup,
            "forward_render_ms": fb.get("forward_render"), "backward_render_ms": fb.get("backward_render"), "phase_ms": phases}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="bicycle_like_6M_1237x822", choices=list(WORKLOADS))
    ap.add_argument("--num-gaussians", type=int, default=0, help="override N (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the short second and third measurements on bicycle_like_6M_survey_c3 and bicycle_like_6M_surface")
    ap.add_argument("--render-frames", type=int, default=10)
    ap.add_argument("--trainer", default="native", choices=["native", "autograd"],
                    help="native: fused HIP activation/Adam around the renderer; autograd: torch.autograd + torch.optim.Adam")
    ap.add_argument("--dense-exchange", action="store_true",
                    help="native trainer: materialise the [N,48] SH gradient and all-reduce it (default: compact exchange + fused SH-grad/Adam)")
    ap.add_argument("--dp-exchange", default="sparse", choices=["sparse", "dense"],
                    help="N>1 (native trainer, compact exchange): sparse = 64-byte records of the Gaussians each view gave a gradient "
                         "to (all-gather of max-count records per rank); dense = all-reduce [N,12] + all-gather [N,3] per view")
    ap.add_argument("--no-dp-side-stream", action="store_true",
                    help="N>1, sparse exchange: do not update the waves no view walked on a side stream under the backward / exchange")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N=1 only: initialise RCCL with world size 1 and issue the data-parallel collectives anyway "
                         "(exercises the N>1 call sequence on a one-GPU box; the number is NOT a bench line)")
    ap.add_argument("--device-pose", dest="host_pose", action="store_false",
                    help="hand the 4x4 camera pose over as a GPU tensor (reference layout; costs one blocking read-back per step)")
    ap.add_argument("--scene-order", action="store_true",
                    help="native trainer: keep the Gaussians in the order the scene generator emits them (random) instead of the "
                         "trainer's default storage order along a Morton curve (NativeGaussianModel(spatial_order=True))")
    ap.add_argument("--full-sort", action="store_true", help="switch GUT_OPT_LAZY_TILE_ORDER off (full 44-bit radix sort)")
    ap.add_argument("--selective-adam", action="store_true", help="visibility-masked Adam (reference SelectiveAdam)")
    ap.add_argument("--fresh-optimizer-state", action="store_true",
                    help="start from all-zero Adam moments (step 0 of a training run) instead of the synthetic mid-training state "
                         "(non-zero moments on every row) the headline is measured in")
    ap.add_argument("--early-extra", type=int, default=None,
                    help="GUT_OPT_EARLY_EXTRA_PERCENT (0..100): share of the row blocks in which the side stream also takes the waves "
                         "with tiles the forward walked nothing of (library default 100)")
    ap.add_argument("--no-placement-tuning", action="store_true",
                    help="keep the trainer state where the allocator first put it (default: NativeTrainStep.tune_placement re-places the three "
                         "[N,48] tensors one at a time, round robin, up to --placement-attempts fresh allocations each, and stops at the first "
                         "move that makes the optimiser's no-op pass > 5 %% faster)")
    ap.add_argument("--placement-attempts", type=int, default=4, help="fresh allocations tried per [N,48] tensor by tune_placement (it stops at the first that is faster)")
    ap.add_argument("--ply", default=None,
                    help="render / train THIS scene instead of the synthetic stand-in: a 3DGS-compatible PLY (threedgrut/model/model.py:"
                         "671-719 layout, 3dgrut_amd/io_ply.py); cameras are the named workload's orbit unless --colmap is given")
    ap.add_argument("--colmap", default=None,
                    help="with --ply: a COLMAP scene directory (sparse/0 + images[_N]/): its training views (poses, intrinsics, images "
                         "as targets where present) replace the orbit cameras (3dgrut_amd/io_colmap.py)")
    ap.add_argument("--colmap-downsample", type=int, default=1, help="images_<N>/ and intrinsics / N (MipNeRF360 runs use 4 for bicycle)")
    ap.add_argument("--no-drop-in", action="store_true",
                    help="skip the short second measurement through the reference's own surface (Tracer.render -> _Autograd -> torch.optim.Adam)")
    ap.add_argument("--no-lazy-moments", action="store_true",
                    help="write both Adam moments of every row every step (default: waves that cannot receive a gradient read their moments, "
                         "bring them up to date in registers and do not write them back; gut_hip.h: GutLazyMoments)")
    ap.add_argument("--force-overlap-optimizer", action="store_true",
                    help="always use the side-stream optimiser pass (default: the trainer alternates the two forms over steps 2-9 and keeps the one "
                         "whose best sample is faster)")
    ap.add_argument("--densification-statistics", action="store_true",
                    help="native trainer: attach strategy.GSStrategy, whose per-view position-gradient statistics run between backward "
                         "and optimiser in every step (the densification phase of a reference run; the number is NOT the headline)")
    ap.add_argument("--no-overlap-optimizer", action="store_true",
                    help="one optimiser kernel after the backward instead of the side-stream pass for the waves that cannot receive a gradient")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing in this process has initialised the GPU yet (device_count() does not, on this image)
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get("GUT_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} started with WORLD_SIZE={world}")
    # GUT_BENCH_SHARE_GPU=1 + GUT_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a box with fewer GPUs than ranks
    # (ranks share the cards, the exchange is host-staged) — exercises launcher, rendezvous and the data-parallel step; the
    # number it prints is NOT a bench line
    share = os.environ.get("GUT_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL prints a version banner on stdout when its first communicator comes up; the contract is ONE JSON line on
    # stdout, so everything before that line is routed to stderr at the file-descriptor level
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if args.force_exchange and world == 1:
        os.environ["GUT_DP_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_PORT", "29531")
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("GUT_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    env = dict(rank=rank, world=world, dev=dev, dist=dist)
    res = run_workload(args, env, args.workload, args.steps, args.warmup, args.render_frames)

    if rank == 0:
        ktimes, stats, stepper = res["ktimes"], res["stats"], res["stepper"]
        W, H, fx = res["W"], res["H"], res["fx"]
        split = ktimes.get("optimizer_early", -1.0) > 0
        bkey = lambda k: "optimizer_late" if (k == "optimizer" and split) else k   # byte model of the kernel as launched
        # dominant kernel = the longest single launch of the step ("optimizer_early" is the span of the side stream's two launches,
        # idle gap included, and is reported under per_kernel only; "optimizer_early_2" is its second launch, timed with events
        # on the side stream it runs on)
        longest = max((k for k in ktimes if ktimes[k] > 0 and k != "optimizer_early"), key=lambda k: ktimes[k])
        # The two compositors are bound by vector-instruction issue, not by bytes or matrix flops: when one of
        # them is the step's longest launch it gets its own block (`roofline_valu`, below) and `roofline` — whose bound is "hbm" or
        # "mfma" by contract — describes the longest launch of the HBM-bound kernels, which is also the one that moves most bytes.
        VALU_BOUND = ("render", "render_bwd")
        dom = max((k for k in ktimes if ktimes[k] > 0 and k != "optimizer_early" and k not in VALU_BOUND), key=lambda k: ktimes[k])
        # achievable HBM bandwidth of THIS box, for context next to the 8 TB/s spec: device-to-device copy of 2 GB
        a = torch.empty(1 << 29, dtype=torch.float32, device=dev); b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize(dev)
        copy_gbs = 4 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        abytes = algorithmic_bytes(stats, bkey(dom), stats["sort_end_bit"])
        achieved = abytes / (ktimes[dom] * 1e-3) / 1e9
        # PMC counters are NOT collected in this run (rocprofv3 --pmc needs its own passes): they are replayed from the
        # tracked profile of the same command and workload, and say so
        prof = profile_counters() if (args.workload == "bicycle_like_6M_1237x822" and not args.num_gaussians and not args.ply
                                      and args.trainer == "native" and not args.scene_order) else {}
        pk = lambda k: prof.get(k, {})
        traffic = pk(dom).get("hbm_bytes")
        roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": (prof.get("_source") + " (separate rocprofv3 --pmc passes of the same workload, replayed here)") if traffic else None,
                    "algorithmic_bytes": abytes, "mean_launch_ms": ktimes[dom], "launches_averaged": res["kcount"],
                    "box_copy_GBps": copy_gbs}
        if pk(dom).get("duration_ms_SQ_A"):
            # the same kernel with nothing beside it: rocprofv3's PMC passes serialise the launches (tracked profile, replayed)
            roofline["alone_in_profile"] = {"ms": pk(dom)["duration_ms_SQ_A"], "hbm_bytes": traffic,
                                            "frac": (traffic or abytes) / (pk(dom)["duration_ms_SQ_A"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "source": prof.get("_source")}
        if dom == "optimizer_early_2":
            roofline["note"] = ("side-stream optimiser pass, second launch: a persistent kernel of two workgroups per CU that streams the "
                                "Adam state of the Gaussians the forward walked nothing of (lazy moment decay: p, m, v read, p and the next "
                                "activation row written) UNDER the VALU-bound backward compositor and then beside the pass over the walked "
                                "waves; it shares the chip by design and is no longer on the step's critical path; alone it runs at the "
                                "box's device-copy rate (profiles/: optimizer_early_2, PMC pass)")
        # (always given, for the longer of the two compositors: which of K7 and the side stream's second launch is the step's
        #  longest launch changes from box to box with the placement of the optimiser state, 0.77 - 0.95 against 0.87 ms)
        roofline_valu = None
        roofline["longest_launch_of_the_step"] = longest + (" (VALU-bound: see roofline_valu)" if longest in VALU_BOUND else "")
        comp = [k for k in VALU_BOUND if ktimes.get(k, -1.0) > 0]
        if comp:
            longest = max(comp, key=lambda k: ktimes[k])
            c = pk(longest)
            if "SQ_INSTS_VALU" in c:
                # wave64 vector instructions per launch (SQ_INSTS_VALU of the tracked PMC pass, same workload) / this run's mean
                # launch duration, against one instruction per SIMD every 2 cycles at 2.4 GHz on 1024 SIMDs (MI355X_MICROARCH.md)
                rate = c["SQ_INSTS_VALU"] / (ktimes[longest] * 1e-3) / 1e9
                roofline_valu = {"kernel": longest, "bound": "valu issue", "achieved": rate, "peak": VALU_PEAK_GINSTR_S,
                                 "unit": "G wave64-instructions/s", "frac": rate / VALU_PEAK_GINSTR_S,
                                 "frac_of_measured_issue_rate": c["SQ_INSTS_VALU"] * VALU_ISSUE_NS * 1e-9 / (1024 * ktimes[longest] * 1e-3),
                                 "wave_instructions_per_launch": c["SQ_INSTS_VALU"], "mean_launch_ms": ktimes[longest],
                                 "counters_source": prof.get("_source"),
                                 "note": "fp32 per-(pixel, Gaussian) arithmetic on the vector pipe (MFMA measured slower, DESIGN.md "
                                         "section 5); the kernel shares its SIMDs with the side-stream optimiser pass while it runs"}
        if split:
            # context for the split optimiser: what the whole Adam step must move vs what of it is left on the critical path
            roofline["optimizer_split"] = {
                "whole_step_algorithmic_bytes": algorithmic_bytes(stats, "optimizer", stats["sort_end_bit"]),
                "critical_path_kernel_bytes": algorithmic_bytes(stats, "optimizer_late", stats["sort_end_bit"]),
                "side_stream_kernel_bytes": algorithmic_bytes(stats, "optimizer_early", stats["sort_end_bit"]),
                "critical_path_ms": ktimes["optimizer"], "side_stream_span_ms": ktimes["optimizer_early"],
                "side_stream_second_launch_ms": ktimes.get("optimizer_early_2")}
        per_kernel = {}
        for k in ktimes:
            e = {"ms": ktimes[k],
                 "GBps": (algorithmic_bytes(stats, bkey(k), stats["sort_end_bit"]) / (ktimes[k] * 1e-3) / 1e9) if ktimes[k] > 0 else None}
            c = pk(k)
            if ktimes[k] > 0 and "SQ_INSTS_VALU" in c:
                # share of the chip's measured wave64 VALU issue rate (1 instruction / 1.16 ns / SIMD, tools/pk_rate.hip):
                # profile's VALU wave-instructions per launch / (this run's duration x 1024 SIMDs x that rate)
                e["valu_frac"] = c["SQ_INSTS_VALU"] * VALU_ISSUE_NS * 1e-9 / (1024 * ktimes[k] * 1e-3)
                e["valu_wave_instructions"] = c["SQ_INSTS_VALU"]
                e["counters_source"] = prof.get("_source")
            if "hbm_bytes" in c:
                e["traffic"] = c["hbm_bytes"]
            if k == "optimizer_early" and ktimes[k] > 0:
                e["note"] = ("side stream, two launches (waves without tiles of the first 60 % of the row blocks under the forward "
                             "compositor, in the kernel's 32-register form; the rest of them plus every wave the forward walked nothing of from the start of the backward "
                             "compositor): ms is the span from the start of the first to the end of the second, idle gap included")
            per_kernel[k] = e
        sh_degree = 3
        out = {
            "metric": "train-step images/sec + render ms/frame, MipNeRF360 bicycle @1/2/4/8 GPU",
            "value": res["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": data_label(args) + (" (REHEARSAL: ranks share GPUs, host-staged exchange - not a bench line)"
                                                      if os.environ.get("GUT_BENCH_SHARE_GPU") == "1" else ""),
            "config": {"workload": args.workload, "num_gaussians": int(stats["num_particles"]), "resolution": [W, H],
                       "sh_degree": sh_degree, "views_per_step": world, "parallelism": f"per-view dp{world}" + ("" if world == 1 else (" dense all-reduce [N,60]" if (args.dense_exchange or args.trainer != "native") else (" sparse exchange: all-gather of 64-byte gradient records" if args.dp_exchange == "sparse" else " all-reduce [N,12] + all-gather [N,3]"))),
                       "exchanged_records_last_step": getattr(stepper, "exchanged_records", None),
                       "loss": "0.8*L1+0.2*(1-SSIM) (HIP fused SSIM)",
                       "optimizer": (("HIP fused SH-gradient+Adam" if not args.dense_exchange else "HIP fused Adam") if args.trainer == "native" else "torch.optim.Adam(fused)") +
                                    (" selective(visibility)" if args.selective_adam else "") + ", all 59 params/Gaussian",
                       "trainer": args.trainer,
                       "optimizer_state": ("all-zero moments (start of training)" if (args.fresh_optimizer_state or args.trainer != "native")
                                           else "synthetic mid-training state: non-zero Adam moments on every row"),
                       "storage_order": ("morton" if (args.trainer == "native" and not args.scene_order) else "as generated"),
                       "placement_trials_ms": [round(t, 3) for t in getattr(stepper, "placement_trials_ms", [])] or None,
                       "optimizer_overlap": bool(getattr(stepper, "overlap_optimizer", False)),
                       "lazy_moment_decay": bool(getattr(stepper, "lazy_moments", False)),
                       "densification_statistics": bool(getattr(stepper, "post_backward_hook", None) is not None),
                       "optimizer_overlap_probe_ms": ({k: round(v, 3) for k, v in stepper._overlap_probe.items() if k in ("ms_on", "ms_off")}
                                                      if getattr(stepper, "_overlap_probe", None) else None)},
            "render_ms_per_frame": res["render_ms"],
            "forward_render_ms_in_train": res["fb"].get("forward_render"), "backward_render_ms_in_train": res["fb"].get("backward_render"),
            "phase_ms": res["phases"], "step_gpu_span_ms": res["step_spans"], "scene_stats": stats, "per_kernel": per_kernel, "roofline": roofline, "roofline_valu": roofline_valu,
            "instrumented_pass": (None if res.get("instrumented_ms") is None else {
                "ms_per_step": res["instrumented_ms"], "steps": args.steps,
                "note": "`value` / `ms_per_step` are timed with no events in the steps (the product's default, enable_kernel_timings = false); "
                        "`per_kernel`, `roofline`, `phase_ms`, `step_gpu_span_ms` and `*_render_ms_in_train` are HIP-event measurements over this "
                        "second pass of the same number of steps, run right after the timed region with every kernel boundary and phase bracketed"}),
            "reference_rtx5090": {"images_per_s": 31.6, "render_ms": 3.64, "note": "README.md:320, different hardware, real dataset"},
        }
        if world > 1 or args.force_exchange:
            # what the communicator looked like from rank 0, so that "did RCCL see N ranks" can be read off the line
            dp_mod = importlib.import_module("3dgrut_amd.dp")
            out["collective"] = dict(dp_mod.collective_info(world), exchange=args.dp_exchange if args.trainer == "native" and not args.dense_exchange else "dense [N,60]",
                                     exchanged_bytes_per_rank=getattr(stepper, "exchanged_bytes_per_rank", None),
                                     exchanged_records_last_step=getattr(stepper, "exchanged_records", None),
                                     record_capacity_overflows=getattr(getattr(stepper, "_exchange", None), "overflows", None),
                                     replica_check="checksums of parameters and moments MIN/MAX-reduced on step 0 (passed, or this line would not exist)",
                                     hardware_status="the N > 1 path has run on multi-GPU hardware only in the driver's own scaling runs; "
                                                     "builder-side it is covered by gloo world-2/3 tests, a world-8 gloo test of the record exchange and two-process one-GPU tests")
        out["config"]["warmup_extended_by"] = res.get("extra_warmup", 0)
        if world == 1 and not args.no_drop_in and args.trainer == "native":
            try:
                out["drop_in"] = run_drop_in(res)
            except Exception as e:
                out["drop_in"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["drop_in"]["selective_adam"] = run_drop_in(res, optimizer_type="selective_adam")
            except Exception as e:
                out["drop_in"]["selective_adam"] = {"error": f"{type(e).__name__}: {e}"}
            try:   # the same with the reference's own activation calls (what `drop_in` was until round 3)
                out["drop_in"]["torch_activations"] = run_drop_in(res, raw_parameters=False)
            except Exception as e:
                out["drop_in"]["torch_activations"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline and res.get("colmap"):
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "not run: the CPU legs are wired for the named workloads' orbit cameras, not for --colmap views"}
        elif world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(res["scene"], res["cams"], res["pose_mod"], W, H, fx, res["c2ws"][0], sh_degree,
                                                   fisheye=res["fisheye"])
                # north_star's wording: a pure-PyTorch per-ray composite on the host cores, same run (second, smaller sample;
                # wired for the pinhole workloads)
                if not res["fisheye"]:
                    out["cpu_baseline_per_ray_torch"] = cpu_baseline_per_ray_torch(res["scene"], res["cams"], res["pose_mod"], W, H, fx,
                                                                                   res["c2ws"][0], sh_degree)
            except Exception as e:  # the baseline is reported, never a gate
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        # workload sensitivity: the same train step on the denser stand-in with SURVEY §8d C3's literal parameters
        # (a short, labelled second measurement; `value` above stays the headline workload)
        if world == 1 and args.workload == "bicycle_like_6M_1237x822" and not args.no_sensitivity and not args.num_gaussians and not args.ply:
            r2 = None
            try:
                del res, stepper
                torch.cuda.empty_cache()
                r2 = run_workload(args, env, "bicycle_like_6M_survey_c3", 10, 5, 5)   # (+ warm-up extended until the overlap probe has decided)
                out["sensitivity"] = {"workload": "bicycle_like_6M_survey_c3",
                                      "note": "same step on the stand-in generated with SURVEY.md §8d C3's literal parameters (log-normal scales "
                                              "mu = ln 0.01, opacity logits N(0, 1.5)): larger, more opaque splats, far more tile intersections",
                                      "value": r2["value"], "unit": "images/s", "ms_per_step": r2["ms_per_step"], "steps": 10, "warmup": 5 + r2.get("extra_warmup", 0),
                                      "render_ms_per_frame": r2["render_ms"], "phase_ms": r2["phases"], "scene_stats": r2["stats"],
                                      "per_kernel_ms": {k: v for k, v in r2["ktimes"].items()}}
            except Exception as e:
                out["sensitivity"] = {"workload": "bicycle_like_6M_survey_c3", "error": f"{type(e).__name__}: {e}"}
            # ... and on the surface-like stand-in, the other end of the bracket around the headline (E/M 0.4 - 0.5 instead of 0.12)
            try:
                r2 = None
                gc.collect()
                torch.cuda.empty_cache()
                r3 = run_workload(args, env, "bicycle_like_6M_surface", 10, 5, 5)
                s3 = r3["stats"]
                out["sensitivity_surface"] = {"workload": "bicycle_like_6M_surface",
                                              "note": "same step on the surface-like stand-in (scenes.scene_surface_like: every Gaussian a flat, mostly "
                                                      "faint disc on a 2-D shell, nothing buried in opaque volumes): about half of every tile's list is "
                                                      "walked, four times the compositing work of the headline stand-in.  Nothing the reference publishes "
                                                      "pins where a trained bicycle sits between the two (unpinned): quote the headline as the range.",
                                              "value": r3["value"], "unit": "images/s", "ms_per_step": r3["ms_per_step"], "steps": 10, "warmup": 5 + r3.get("extra_warmup", 0),
                                              "render_ms_per_frame": r3["render_ms"], "phase_ms": r3["phases"], "scene_stats": s3,
                                              "E_over_M": s3["traversed_fwd"] / max(1, s3["num_intersections"]), "M_over_V": s3["num_intersections"] / max(1, s3["num_visible"]),
                                              "optimizer_overlap": bool(getattr(r3["stepper"], "overlap_optimizer", False)),
                                              "per_kernel_ms": {k: v for k, v in r3["ktimes"].items()}}
                out["headline_range_images_per_s"] = sorted([out["sensitivity_surface"]["value"], out["value"]])
            except Exception as e:
                out["sensitivity_surface"] = {"workload": "bicycle_like_6M_surface", "error": f"{type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
