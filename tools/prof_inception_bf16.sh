#!/bin/bash
# rocprofv3 kernel trace of the Inception-v3 bf16 Fast R-CNN bench: per-kernel totals of the last image -> stdout
cd /tmp && export TMPDIR=/tmp
mkdir -p /tmp/prof_in
rocprofv3 --kernel-trace --stats -d /tmp/prof_in -o in --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_inception.py 2000 $1 > /tmp/prof_in/log.txt 2>&1
t=$(find /tmp/prof_in -name '*kernel_trace.csv' | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "mpn::" in r["Kernel_Name"] and "pack_" not in r["Kernel_Name"]]
# last image: from the last image_transform kernel on
idx = max(i for i, r in enumerate(sel) if "image_transform" in r["Kernel_Name"])
img = sel[idx:]
tot = collections.OrderedDict()
for r in img:
    k = r["Kernel_Name"].split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    c = tot.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += d
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:10.1f} us  {n:4d}  {k}")
print("total", sum(d for _, d in tot.values()))
PY
