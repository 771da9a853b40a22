#!/usr/bin/env python3
"""per-kernel sums of a rocprofv3 --pmc counter_collection.csv (kernels of this library only)"""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))[-1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "anonymous namespace" not in name or "at::native" in name:
        continue
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")[:44]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[name].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
print("kernel".ljust(46) + "calls " + " ".join(n.replace("SQ_", "")[:14].rjust(15) for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(calls[k])
    print(k.ljust(46) + f"{n:5d} " + " ".join(f"{v.get(c, 0) / n:15.4g}" for c in names))
