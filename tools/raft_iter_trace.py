#!/usr/bin/env python3
"""Per-launch durations of ONE RAFT refinement iteration, from a rocprofv3 --kernel-trace CSV of tools/raft_bench.py:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python3 $GRAFT_REPO_ROOT/tools/raft_bench.py 31
   python3 tools/raft_iter_trace.py /tmp/rt"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# an iteration starts at each lookup launch; take one from the middle of the last timed pass
idx = [i for i, n in enumerate(names) if "raft_corr_lookup" in n or "raft_lookup_convc1" in n]
a = idx[-10]; b = idx[-9]
tot = 0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{d:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  {r['Kernel_Name'][:110]}")
gap = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print(f"sum {tot:.1f} us, wall {gap:.1f} us")
