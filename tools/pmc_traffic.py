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
    if "gemm_bf16" not in k and "attn_bf16" not in k and "conv3x3_c64" not in k and "gru_half_kernel" not in k and "stem7x7" not in k and "conv_h8" not in k:
        continue
    if want == "conv" and ", true, " not in k and "conv3x3_c64" not in k and "gru_half_kernel" not in k and "stem7x7" not in k and "conv_h8" not in k:
        continue
    f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
    res[k] = {"launches": len(fe[k]), "FETCH_SIZE_KiB_avg": round(f, 1), "WRITE_SIZE_KiB_avg": round(w, 1),
              "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
# the update block's memory-side kernels beside the conv family (not read by bench.py): raw counters; the x2 read correction is established for wide
# coalesced streams only, so the gather kernel's corrected figure is an upper bound
mem = {}
if want == "conv":
    for k in fe:
        if any(t in k for t in ("lookup_convc1", "raft_convf1", "norm_apply", "corr_lookup", "flow_head2")):
            f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
            mem[k] = {"launches": len(fe[k]), "FETCH_SIZE_KiB_avg": round(f, 1), "WRITE_SIZE_KiB_avg": round(w, 1),
                      "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
doc = {"note": "FETCH_SIZE doubled (gfx950 correction for 16 B/lane coalesced reads), WRITE_SIZE as read; KiB -> bytes", "kernels": res}
if mem:
    doc["memory_kernels"] = mem
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
