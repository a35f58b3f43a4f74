#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv directory: per kernel name, mean of each counter per dispatch."""
import csv, glob, os, sys, collections
src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "gemm" not in k and "attn" not in k and "gru_half" not in k and "conv3x3" not in k and "conv_h8" not in k and (len(sys.argv) < 3):
        continue
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
