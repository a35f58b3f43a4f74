#!/usr/bin/env python3
"""Sum one PMC counter per kernel name from a rocprofv3 --pmc CSV directory: usage pmc_kernel.py DIR [substring]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, set()])
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:70], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r["Dispatch_Id"])
for (k, c), (v, d) in sorted(acc.items()):
    print(f"{k:70s} {c:12s} total {v:.4g}  dispatches {len(d)}  per dispatch {v / max(len(d), 1):.4g}")
