#!/usr/bin/env python3
"""Per kernel and per loop (backward branch) instruction mix of a .hip file compiled for gfx950 -- the tool the round-4 epilogue /
attention / GRU findings came from (selects for every element of an unrolled loop, SGPR spills as v_readlane, per-element scalar tests,
scratch reloads behind vmcnt(0)).   python tools/exp/isa_scan.py videotgb_amd/csrc/gemm_pp.hip [name substring]"""
import collections, os, re, subprocess, sys
src = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/isa_scan.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-function", "-I", repo + "/include",
                "-I", repo + "/videotgb_amd/csrc", "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
s = open(out).read().split("\n")
def kind(op):
    if "mfma" in op: return "mfma"
    if "readlane" in op or "writelane" in op: return "lane"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("v_"): return "valu"
    if "cbranch" in op or op == "s_branch": return "branch"
    if op == "s_nop": return "nop"
    if op == "s_waitcnt": return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return "other"
for st in [i for i, l in enumerate(s) if re.match(r"^_Z\w+:", l)]:
    name = s[st].split(":")[0]
    if want not in name: continue
    en = next(i for i in range(st, len(s)) if "s_endpgm" in s[i])
    lines = [l.strip() for l in s[st:en] if l.strip() and not l.strip().startswith(";")]
    lines = [l for l in lines if not (l.startswith(".") and not re.match(r"^\.LBB", l))]
    tot = collections.Counter(kind(x.split()[0]) for x in lines if not x.startswith(".LBB"))
    print(f"{name[:100]}\n   total {len(lines)}: {dict(tot)}")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    seen = set()
    for i, l in enumerate(lines):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (labels[m.group(1)], i) not in seen:
            seen.add((labels[m.group(1)], i))
            seg = [x for x in lines[labels[m.group(1)]:i] if not x.startswith(".LBB")]
            if len(seg) >= 40:
                c = collections.Counter(kind(x.split()[0]) for x in seg)
                top = collections.Counter(x.split()[0] for x in seg if x.startswith("v_") and "mfma" not in x).most_common(6)
                print(f"   loop {len(seg):5d}: {dict(c)}  top valu {top}")
