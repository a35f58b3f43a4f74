#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE reports half of a wide
coalesced stream; both counters are in KiB)."""
import csv, glob, json, os, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
want = sys.argv[4] if len(sys.argv) > 4 else ""      # "conv": the implicit-GEMM instantiations (<EPI, 0, true, NWN>) only
def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
res = {}
for k in fe:
    if "gemm_bf16" not in k and "attn_bf16" not in k and "conv3x3_c64" not in k and "gru_half_kernel" not in k and "stem7x7" not in k:
        continue
    if want == "conv" and ", true, " not in k and "conv3x3_c64" not in k and "gru_half_kernel" not in k and "stem7x7" not in k:
        continue
    f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
    res[k] = {"launches": len(fe[k]), "FETCH_SIZE_KiB_avg": round(f, 1), "WRITE_SIZE_KiB_avg": round(w, 1),
              "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
json.dump({"note": "FETCH_SIZE doubled (gfx950 correction for 16 B/lane coalesced reads), WRITE_SIZE as read; KiB -> bytes", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
