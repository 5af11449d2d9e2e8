"""Dev tool: turn the output of tools/collect_profiles.sh (gpurun_out/profiles) into profiles/<round>/:
bench_kernel_stats.csv, bench_under_rocprof.json, bench_plain.json, pmc_traffic.json (HBM bytes per launch),
pmc_sq.json (SQ / TCC / GRBM counters per launch + derived VALU fractions), README.md."""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "round3"
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
# (regex on the demangled kernel name, bench key).  Templated names: k_render<true>(, k_render_backward<false>(, k_sh_adam<true>(
KEYS = [(r"\bk_project_on_tiles\b", "project"), (r"\bk_scan_wave_sums\b", "scan"), (r"\bk_expand_tiles\b", "expand"), (r"\bk_tile_ranges\b", "ranges"),
        (r"\bk_render(<[^>]*>)?\(", "render"), (r"\bk_render_backward\b", "render_bwd"), (r"\bk_project_backward", "project_bwd"),
        (r"\bk_sh_adam\b", "optimizer"), (r"\bk_adam_rows_without_gradient_narrow\b", "optimizer_early_narrow"),
        (r"\bk_adam_rows_without_gradient\b", "optimizer_early"), (r"onesweep|radix_sort|OneSweep", "sort"),
        (r"\bk_ssim_|k_photometric", "loss")]
VALU_ISSUE_NS = 1.16      # one wave64 v_fma_f32 per SIMD every 1.16 ns with >= 2 waves resident (tools/pk_rate.hip, measured)
SIMDS = 1024


def key_of(name):
    for pat, v in KEYS:
        if re.search(pat, name):
            return v
    return None


PMC_STEPS = 10   # tools/collect_profiles.sh: fwd_once.py 6000000 10 step


def _second_half(rows_by_name):
    """fwd_once runs PMC_STEPS train steps over the bench's views: keep each kernel's dispatches of the LAST step — the steady
    state (every view seen once, so the Adam moments of the rows are what they are in the bench; and the first dispatch of a
    kernel with a spill area — K6 — can include the runtime's one-off scratch set-up, 0.54 -> 1.45 ms)."""
    out = []
    kn = lambda name: name[0] if isinstance(name, tuple) else name
    # the side-stream kernel has two instantiations: <true> (lazy moment decay: the train steps) and <false> (tune_placement's no-op
    # passes, trainers that write their moments every step); when both ran, the steps are the <true> ones
    lazy_steps = any(re.search(r"\bk_adam_rows_without_gradient<true>", kn(name)) for name in rows_by_name)
    # since the end of round 4 the step's first side-stream launch is the kernel's 32-register form (its own name) and only the
    # second one is k_adam_rows_without_gradient<true>: one dispatch of each per step
    narrow = any(re.search(r"\bk_adam_rows_without_gradient_narrow\b", kn(name)) for name in rows_by_name)
    for name, rows in rows_by_name.items():
        kname = kn(name)
        if lazy_steps and re.search(r"\bk_adam_rows_without_gradient<false>", kname):
            continue
        if re.search(r"\bk_adam_rows_without_gradient_narrow\b", kname):
            out.extend(rows[-1:])
            continue
        if re.search(r"\bk_adam_rows_without_gradient\b", kname):
            out.extend(rows[-1:] if narrow else rows[-2:])   # (the run also holds the no-op passes of tune_placement)
            continue
        per_step = max(1, len(rows) // PMC_STEPS)
        out.extend(rows[-per_step:] if len(rows) >= PMC_STEPS else rows[len(rows) // 2:])
    return out


def counters(pass_name):
    files = glob.glob(os.path.join(src, "pmc_" + pass_name, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    by = {}
    for r in csv.DictReader(open(files[0])):
        by.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(r)
    acc = {}
    for r in _second_half(by):
        k = key_of(r["Kernel_Name"])
        if k:
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    # the side-stream optimiser kernel is launched twice per step: report the step's total under optimizer_early and the second
    # launch (the long one, under the backward compositor and after it) under optimizer_early_2
    if "optimizer_early" in acc and all(len(v) == 2 for v in acc["optimizer_early"].values()):
        out["optimizer_early"] = {c: sum(v) for c, v in acc["optimizer_early"].items()}
        out["optimizer_early_2"] = {c: v[1] for c, v in acc["optimizer_early"].items()}
    elif "optimizer_early" in acc and "optimizer_early_narrow" in acc:
        out["optimizer_early_2"] = {c: v[-1] for c, v in acc["optimizer_early"].items()}
        out["optimizer_early"] = {c: v[-1] + acc["optimizer_early_narrow"][c][-1] for c, v in acc["optimizer_early"].items()
                                  if c in acc["optimizer_early_narrow"]}
    return out


def kernel_durations(pass_name):
    files = glob.glob(os.path.join(src, "pmc_" + pass_name, "**", "*kernel_trace.csv"), recursive=True)
    acc = {}
    if files:
        by = {}
        for r in csv.DictReader(open(files[0])):
            by.setdefault(r["Kernel_Name"], []).append(r)
        for r in _second_half(by):
            k = key_of(r["Kernel_Name"])
            if k:
                acc.setdefault(k, []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    if len(acc.get("optimizer_early", [])) == 2:
        out["optimizer_early"] = sum(acc["optimizer_early"])
        out["optimizer_early_2"] = acc["optimizer_early"][1]
    elif acc.get("optimizer_early") and acc.get("optimizer_early_narrow"):
        out["optimizer_early_2"] = acc["optimizer_early"][-1]
        out["optimizer_early"] = acc["optimizer_early"][-1] + acc["optimizer_early_narrow"][-1]
    return out


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(dst, "bench_kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "bench_plain.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))

traffic = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over tools/fwd_once.py 6000000 10 step (ten full "
                   "native train steps over the bench views, two-pass optimiser forced on, bicycle-like stand-in, 1 MI355X), last step, mean per launch.  FETCH_SIZE/WRITE_SIZE are in KiB.  Per "
                   "MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of the bytes of wide (16 B/lane) streaming reads: "
                   "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.  The factor 2 is calibrated for 16 B/lane streams only; for the "
                   "gather-heavy kernels (render, render_bwd: 48-byte row gathers) hbm_bytes_uncorrected = (FETCH+WRITE)*1024 is the "
                   "lower bound and hbm_bytes the upper bound.",
           "kernels": {}}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, d in counters(ctr).items():
        if ctr in d:
            traffic["kernels"].setdefault(k, {})[ctr + "_KiB"] = d[ctr]
for k, d in traffic["kernels"].items():
    f, w = d.get("FETCH_SIZE_KiB", 0.0), d.get("WRITE_SIZE_KiB", 0.0)
    d["hbm_bytes"] = 2 * f * 1024 + w * 1024
    d["hbm_bytes_uncorrected"] = (f + w) * 1024
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

sq = {"note": "separate rocprofv3 --pmc passes (SQ_A, SQ_B, TCC) over tools/fwd_once.py 6000000 10 step, mean per launch, values summed over "
              "the chip as rocprofv3 reports them.  duration_ms is the kernel-trace duration in the SAME (profiled) pass; every figure is the mean over the dispatches of the last of the ten steps.  "
              f"valu_issue_frac = SQ_INSTS_VALU x {VALU_ISSUE_NS} ns / ({SIMDS} SIMDs x duration): share of the chip's measured wave64 VALU "
              "issue rate (tools/pk_rate.hip) the kernel's vector instructions account for.  valu_active_frac = 4 x SQ_ACTIVE_INST_VALU / "
              "({SIMDS} SIMDs x GRBM_GUI_ACTIVE / 8): SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles, GRBM_GUI_ACTIVE is summed "
              "over the 8 XCDs (MI355X_MICROARCH.md, cycle constants and DVFS section); ~1.0 means the vector pipes never idle.  "
              "wave_wait_share / wave_issue_stall_share = SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES.  "
              "atomic_GBps = TCC_EA0_ATOMIC_sum x 64 B / duration (chip-wide float-atomic rate is about 1300 GB/s).",
      "kernels": {}}
for p in ("SQ_A", "SQ_B", "TCC"):
    dur = kernel_durations(p)
    for k, d in counters(p).items():
        e = sq["kernels"].setdefault(k, {})
        e.update(d)
        if k in dur:
            e["duration_ms_" + p] = dur[k]
for k, e in sq["kernels"].items():
    if "SQ_INSTS_VALU" in e and "duration_ms_SQ_A" in e:
        e["valu_issue_frac"] = e["SQ_INSTS_VALU"] * VALU_ISSUE_NS * 1e-9 / (SIMDS * e["duration_ms_SQ_A"] * 1e-3)
    if e.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in e:
        e["valu_active_frac"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / (SIMDS * e["GRBM_GUI_ACTIVE"] / 8.0)
        e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (e["duration_ms_SQ_A"] * 1e-3) / 1e9
    if e.get("SQ_WAVE_CYCLES") and e.get("SQ_WAIT_ANY") is not None and e.get("SQ_WAVES"):
        # SQ_A and SQ_B are separate passes of the same work: shares of a wave's lifetime
        e["wave_wait_share"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        e["wave_issue_stall_share"] = e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    if "TCC_EA0_ATOMIC_sum" in e and "duration_ms_TCC" in e:
        e["atomic_GBps"] = e["TCC_EA0_ATOMIC_sum"] * 64 / (e["duration_ms_TCC"] * 1e-3) / 1e9
    if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_ANY" in e:
        pass
json.dump(sq, open(os.path.join(dst, "pmc_sq.json"), "w"), indent=1)
print(json.dumps({k: {c: v for c, v in e.items() if "frac" in c or "GBps" in c or "duration" in c} for k, e in sq["kernels"].items()}, indent=1))
print(json.dumps(traffic["kernels"], indent=1))

# ---- README.md of the profile directory ----
rows = list(csv.DictReader(open(os.path.join(dst, "bench_kernel_stats.csv"))))
plain = json.load(open(os.path.join(dst, "bench_plain.json")))
prof = json.load(open(os.path.join(dst, "bench_under_rocprof.json")))
L = ["# Profiles (" + rnd + ")\n",
     "Collected with `tools/collect_profiles.sh` on one MI355X (gpurun), post-processed by `tools/profile_report.py`.\n",
     "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-sensitivity` (the plain run next to it: `python3 bench.py --steps 20 --warmup 5`)\n",
     "Files: `bench_kernel_stats.csv` (rocprofv3 per-kernel summary of that run), `bench_under_rocprof.json` (the bench line printed "
     "under the profiler), `bench_plain.json` (same command without the profiler, same box), `pmc_traffic.json` (HBM bytes per launch "
     "from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes), `pmc_sq.json` (SQ / TCC / GRBM counters per launch from three more "
     "separate `--pmc` passes, with the derived VALU fractions; formulas in its `note`).  All PMC passes run "
     "`tools/fwd_once.py 6000000 10 step` (ten full native train steps over the bench's 8 views, two-pass optimiser forced on of the bench scene).\n",
     f"Bench line (plain): **{plain['value']:.1f} images/s, {plain['ms_per_step']:.3f} ms/step**, forward render "
     f"{plain['render_ms_per_frame']:.3f} ms/frame; under the profiler {prof['value']:.1f} images/s.  Phases (HIP events, ms): "
     + ", ".join(f"{k} {v:.3f}" for k, v in plain["phase_ms"].items()) + ".\n",
     f"Roofline block of the plain run: dominant kernel `{plain['roofline']['kernel']}`, {plain['roofline']['achieved']:.0f} GB/s of "
     f"algorithmic bytes = {plain['roofline']['frac']:.2f} of the 8 TB/s spec; a 2 GB device copy on the same box ran at "
     f"{plain['roofline'].get('box_copy_GBps', float('nan')):.0f} GB/s.\n",
     "Per-kernel counters (per launch):\n",
     "| kernel | duration ms (profiled) | HBM bytes (2xFETCH+WRITE) | uncorrected | VALU wave-instr | valu_issue_frac | valu_active_frac | LDS instr | bank-conflict cycles | atomic GB/s |\n|---|---|---|---|---|---|---|---|---|---|"]
for k in ("project", "expand", "sort", "render", "loss", "render_bwd", "optimizer", "optimizer_early", "optimizer_early_2"):
    e, t = sq["kernels"].get(k, {}), traffic["kernels"].get(k, {})
    if not e and not t:
        continue
    g = lambda d, c, f="{:.3g}": f.format(d[c]) if c in d else "-"
    L.append(f"| {k} | {g(e, 'duration_ms_SQ_A', '{:.3f}')} | {g(t, 'hbm_bytes', '{:.3e}')} | {g(t, 'hbm_bytes_uncorrected', '{:.3e}')} | "
             f"{g(e, 'SQ_INSTS_VALU', '{:.3e}')} | {g(e, 'valu_issue_frac', '{:.2f}')} | {g(e, 'valu_active_frac', '{:.2f}')} | "
             f"{g(e, 'SQ_INSTS_LDS', '{:.3e}')} | {g(e, 'SQ_LDS_BANK_CONFLICT', '{:.3e}')} | {g(e, 'atomic_GBps', '{:.0f}')} |")
L += ["", "`optimizer_early` is the side-stream optimiser pass, two launches per step (beside K6 the kernel's 32-register form "
          "`k_adam_rows_without_gradient_narrow`, from K7's start `k_adam_rows_without_gradient<true>`): its row is the step's total (both "
          "launches, run alone here because the PMC passes serialise the kernels), `optimizer_early_2` the second launch.",
      "", "Top kernels (all launches of the run: warm-up, the 20 timed and the 20 instrumented train steps, 10 render-only frames, the drop-in legs, setup):\n",
      "| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|"]
for r in rows[:18]:
    L.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
L.append("")
dom_key = plain["roofline"]["kernel"]
second_launch = dom_key == "optimizer_early_2"   # same kernel as optimizer_early: the second of its two launches per step
dom = [r for r in rows if key_of(r["Name"]) == ("optimizer_early" if second_launch else dom_key)]
if dom:
    # the same kernel's dispatches inside the bench's TIMED region: the last `steps` dispatches in the kernel trace of the stats
    # run (the run-wide average above also holds the warm-up, whose steps 1-4 alternate the one- and two-pass optimiser)
    tr = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)
    timed = None
    if tr:
        d = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(tr[0]))
             if key_of(r["Kernel_Name"]) == ("optimizer_early" if second_launch else dom_key)]
        w0, k0 = int(prof["warmup"]), int(prof["steps"])
        narrow_first = any(key_of(r["Kernel_Name"]) == "optimizer_early_narrow" for r in csv.DictReader(open(tr[0])))
        if second_launch and narrow_first:
            d = d[-k0:]             # the first launch is a kernel of its own: one dispatch of this one per step
        elif second_launch:
            d = d[-2 * k0:][1::2]   # the stats run ends with the timed steps (--no-sensitivity): two launches each, the second
        else:
            d = d[w0:w0 + k0]       # the headline workload comes first in the run
        timed = sum(d) / len(d) if d else None
    L.append(f"rocprofv3's average for the dominant kernel (`{dom[0]['Name'][:40]}`) over the whole run: {float(dom[0]['AverageNs']) / 1e3:.1f} us"
             + (f"; over the dispatches of the bench's timed region: **{timed:.1f} us**" if timed else "")
             + f".  Live hipEvent mean in the bench line of the same run (`roofline.mean_launch_ms`): {prof['roofline']['mean_launch_ms'] * 1e3:.1f} us "
             f"(plain run: {plain['roofline']['mean_launch_ms'] * 1e3:.1f} us).\n")
open(os.path.join(dst, "README.md"), "w").write("\n".join(L))
