#!/bin/bash
# rocprofv3 kernel trace of the single-clip step (bench.py --clips 1): where the batch-1 latency goes.  usage: tools/single_clip_prof.sh [tag] [extra bench args]
TAG=${1:-r05_single_clip}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sc_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --clips 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-prof "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}.err
cd $GRAFT_REPO_ROOT
python3 tools/summarize_rocprof.py /tmp/sc_$TAG gpurun_out/${TAG}_kernel_stats bench.py --clips 1 --steps 10 --warmup 3 "$@" > /dev/null
python3 - <<PY
import csv, glob, json
d = json.loads(open("gpurun_out/${TAG}.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
f = glob.glob("/tmp/sc_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last raft_stem / first kernel of a step ... use the final 1/13 of the kernels by time
t_end = int(rows[-1]["End_Timestamp"]); span = d["ms_per_step"] * 1e6
last = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - span]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print(f"last step window: {len(last)} launches, kernel time {busy / 1e6:.2f} ms of {span / 1e6:.2f} ms wall -> gaps {(span - busy) / 1e6:.2f} ms")
PY
head -45 gpurun_out/${TAG}_kernel_stats.md
