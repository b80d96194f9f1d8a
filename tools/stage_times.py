#!/usr/bin/env python
"""Developer tool: per-stage ms of one bench step (bench.py JSON on stdin or run inline)."""
import json, subprocess, sys, os
env = dict(os.environ)
out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True, env=env)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
except Exception:
    print(out.stdout[-2000:], out.stderr[-3000:]); sys.exit(1)
print(os.environ.get("DERP_LIB", "default"), "ms/step", d["ms_per_step"], "value", d["value"],
      {k: v for k, v in d["stage_ms_per_step"].items() if v > 1.0}, "pp_l0_ms", d["roofline"]["kernel_ms"], "alg_GB", round(d["roofline"]["logical_gathers"]["bytes_per_launch_executed"] / 1e9, 1), "n_cost_M", round(d["roofline"]["logical_gathers"]["n_cost_per_launch"] / 1e6, 1))
