#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gpu_suite_final.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gpu_suite_final.txt
tail -8 gpurun_out/r06_gpu_suite_final.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source','')[-48:], d['cpu_baseline']['value'], json.dumps(d.get('fc_split3_aux'))[:200])
PY
