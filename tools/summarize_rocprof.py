#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` directory into the small
markdown/JSON summary committed under profiles/ (per-kernel calls, total, average, share)."""
import csv
import glob
import json
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_stats.csv under {src}")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    agg = {}
    for r in rows:
        name = r.get("Name") or r.get("KernelName")
        calls = int(r.get("Calls", 0))
        total = float(r.get("TotalDurationNs", 0))
        a = agg.setdefault(name, [0, 0.0])
        a[0] += calls
        a[1] += total
    tot = sum(v[1] for v in agg.values()) or 1.0
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = [{"kernel": k, "calls": c, "total_ms": round(t / 1e6, 3), "avg_us": round(t / max(c, 1) / 1e3, 2),
            "share": round(t / tot, 4)} for k, (c, t) in items]
    with open(dst + ".json", "w") as fh:
        json.dump({"source": "rocprofv3 --kernel-trace --stats", "command": " ".join(sys.argv[3:]), "kernels": out[:60]}, fh, indent=1)
    with open(dst + ".md", "w") as fh:
        fh.write(f"# rocprofv3 kernel stats\n\ncommand: `{' '.join(sys.argv[3:])}`\n\n| kernel | calls | total ms | avg us | share |\n|---|---|---|---|---|\n")
        for o in out[:40]:
            fh.write(f"| `{o['kernel'][:110]}` | {o['calls']} | {o['total_ms']} | {o['avg_us']} | {o['share'] * 100:.1f}% |\n")
    print(open(dst + ".md").read()[:3000])


if __name__ == "__main__":
    main()
