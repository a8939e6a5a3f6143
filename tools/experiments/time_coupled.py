#!/usr/bin/env python
"""Stage times of the benchmark pair (phantom and zero background) with the per-stage events: python tools/experiments/time_coupled.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import subprocess
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"], stdout=subprocess.PIPE, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print("value %.1f zero_bg %.1f coupled %s" % (d["value"], d["value_zero_background"], {k: round(v, 4) for k, v in d["coupled_convex_ms"].items() if isinstance(v, float)}))
print({k: round(v, 4) for k, v in d["stages_ms"].items()})
