#!/bin/bash
# Per-launch timeline of ONE image of a bench configuration (rocprofv3 --kernel-trace; serialised by the profiler, so the
# durations are the kernels' own, the gaps are not the pipeline's).  usage: tools/kernel_timeline.sh c1 [extra bench args]
cfg=${1:-c1}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline "$@" > /tmp/kt_bench.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "image_transform" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
tot = 0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    tot += d
    print("%8.1f us  +%8.1f  %-64s grid %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1000, r["Kernel_Name"][:64], r.get("Grid_Size_X", "")))
print("sum of kernel durations %.1f us" % tot)
PY
