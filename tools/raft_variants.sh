#!/bin/bash
# GPU box: per-kernel-variant average times of the RAFT stage (B clips; NWN = forced conv tile: 0 default, 3 = 256x128 instead of 512x128; needs a VTGB_DEBUG_HOOKS build)
B=${1:-31}; NWN=${2:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt -- python3 $GRAFT_REPO_ROOT/tools/raft_bench.py $B $NWN > /tmp/rt.log 2>&1
tail -1 /tmp/rt.log
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/rt/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:78]:78s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
