#!/bin/bash
# rocprofv3 kernel stats + per-launch timeline of tools/bench_resnet.py (args: its args; default "50 1000 bf16") -> gpurun_out/prof_rn/
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_rn
mkdir -p /tmp/prof_rn
rocprofv3 --kernel-trace --stats -d /tmp/prof_rn -o rn --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_resnet.py ${@:-50 1000 bf16} > /tmp/prof_rn/log.txt 2>&1
f=$(find /tmp/prof_rn -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_rn/kernel_stats.csv
head -12 "$f" | cut -c1-200
t=$(find /tmp/prof_rn -name '*kernel_trace.csv' | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "mpn::" in r["Kernel_Name"] and "pack_" not in r["Kernel_Name"]]
n = len(sel) // 6
for r in sel[-n:]:
    print(r["Kernel_Name"][5:35], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
PY
