"""Condensed view of a bench.py JSON line read from stdin."""
import json
import sys

d = json.loads(sys.stdin.read())
c, r = d["config"], d["roofline"]
print(d["steps"], d["warmup"], c.get("mode"), c["scenarios_per_gpu"], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3),
      "conv", round(c["converged_last_step"], 4), "ipm iters", c.get("ipm_iterations_rank0"), "kernel ms", round(r["kernel_ms_per_launch"], 2),
      "cpu", d.get("cpu_baseline", {}).get("value"))
