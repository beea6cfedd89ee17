#!/usr/bin/env python3
"""One line per bench.py JSON file: ms per step, iterations, Arnoldi launch time, roofline fraction.  Usage: bench_ab_summary.py FILE ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        its = d["config"].get("gmres_iterations_per_step")
        print(f, round(d["ms_per_step"], 1), its if len(its) <= 6 else its[:3] + ["..."] + its[-2:],
              round(r.get("avg_launch_us", 0), 2), r.get("launches"), round(r.get("frac", 0), 4))
    except Exception as e:
        print(f, "ERR", e)
