#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: average counter value per kernel name.  Usage: pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void npg::", "").replace("npg::", "")
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    print(f"== {d}")
    for name in sorted(acc):
        for c, (s, n) in acc[name].items():
            print(f"{name:40s} {c:12s} launches {n:6d}  avg {s / n:14.1f}")
