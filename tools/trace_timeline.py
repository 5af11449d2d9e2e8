"""Dev tool: timeline of ONE train step from a rocprofv3 --kernel-trace CSV (per-dispatch start / end timestamps): every dispatch of the
last complete step with its queue, start offset, duration and the idle gap to the previous dispatch on the same queue.
usage: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [step_from_end]"""
import csv, glob, re, sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:60]
# a step starts at a k_project_on_tiles dispatch that is followed (before the next one) by a k_render_backward
starts = [i for i, r in enumerate(rows) if "k_project_on_tiles" in r["Kernel_Name"]]
steps = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    if any("k_render_backward" in r["Kernel_Name"] for r in rows[a:b]):
        steps.append((a, b))
a, b = steps[-back]
t0 = int(rows[a]["Start_Timestamp"])
last_end = {}
print(f"{f}: step {len(steps) - back} of {len(steps)}, {b - a} dispatches, span {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us to the last dispatch's end")
tot_gap = {}
for r in rows[a:b]:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    tot_gap[q] = tot_gap.get(q, 0.0) + max(gap, 0.0)
    print(f"q{q:>3} +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {short(r['Kernel_Name'])}")
print("idle between dispatches per queue (us):", {k: round(v, 1) for k, v in tot_gap.items()})
nxt = int(rows[b]["Start_Timestamp"]) if b < len(rows) else None
if nxt:
    print(f"next step's first dispatch at +{(nxt - t0) / 1e3:.1f} us")
