#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/r06_split3_gate.sh
grep -v amdgpu gpurun_out/r06_split3_gate.txt | grep "FAILED\|^E  " | head -20 | cut -c1-300
python bench.py --config c3 --fc-arith split3 --steps 8 --warmup 3 > gpurun_out/r06_bench_c3_split3.json 2> gpurun_out/r06_bench_c3_split3.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06_bench_c3_split3.json').read().strip().splitlines()[-1])
    print("c3 split3:", d['value'], d['ms_per_step'], json.dumps({k:v.get('ms_per_image') for k,v in d['kernels'].items()}))
except Exception as e:
    print("failed", e); print(open('gpurun_out/r06_bench_c3_split3.err').read()[-1200:])
PY
