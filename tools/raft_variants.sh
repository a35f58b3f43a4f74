#!/bin/bash
# GPU box: per-kernel-variant average times of the RAFT stage (B = 8 clips)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt -- python3 $GRAFT_REPO_ROOT/tools/raft_bench.py 8 > /tmp/rt.log 2>&1
tail -1 /tmp/rt.log
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/rt/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
PY
