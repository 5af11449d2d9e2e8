"""Dev tool: one-line summary of bench.py JSON outputs."""
import json, sys
for f in sys.argv[1:]:
    j = json.load(open(f))
    print(f, round(j["value"], 1), "img/s", round(j["ms_per_step"], 3), "ms/step  render", round(j.get("render_ms_per_frame", 0), 3),
          {k: round(v["ms"], 3) for k, v in j["per_kernel"].items()}, {k: round(v, 3) for k, v in j.get("phase_ms", {}).items()})
