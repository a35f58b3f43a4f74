#!/usr/bin/env python3
"""Per-call-site breakdown of the RAFT stage from a `rocprofv3 --kernel-trace --output-format csv`
directory: dispatches are keyed by (kernel, grid, LDS bytes, ordinal since the last corr lookup), which
separates the convolutions of one refinement iteration although several share a kernel instantiation.
usage: raft_trace.py <rocprof dir> <out.md>"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name[:70]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {src}")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg, seen = {}, {}
    for r in rows:
        nm = short(r["Kernel_Name"])
        if "corr_lookup" in nm:
            seen = {}
        base = (nm, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", ""))
        seen[base] = seen.get(base, 0) + 1
        key = base + (seen[base],)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values()) or 1.0
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w") as fh:
        fh.write("| kernel | grid | lds | ordinal | calls | total ms | avg us | share |\n|---|---|---|---|---|---|---|---|\n")
        for (nm, g, l, o), (c, t) in items[:60]:
            fh.write(f"| `{nm}` | {g} | {l} | {o} | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {t / tot * 100:.1f}% |\n")
    print(open(dst).read()[:6000])


if __name__ == "__main__":
    main()
