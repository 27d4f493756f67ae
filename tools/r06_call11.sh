#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MPN_FLAVOUR=debug python tools/hook_ab.py c3 12 base mpn_pool_knock=1 > gpurun_out/r06_c3_pool_knock.txt 2>&1; grep -v amdgpu gpurun_out/r06_c3_pool_knock.txt
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source','')[-60:], json.dumps(d.get('fc_split3_aux')))
PY
