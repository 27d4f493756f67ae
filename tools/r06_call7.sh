#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -m gpu -q -s -p no:cacheprovider -k "three_plane" > gpurun_out/r06_call7_split3.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call7_split3.txt
grep -v amdgpu gpurun_out/r06_call7_split3.txt | grep "N = \|float64\|passed\|failed\|rc \|^E " | cut -c1-400
bash tools/r06_split3_gate.sh
python bench.py --fc-arith split3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_split3.json 2> gpurun_out/r06_bench_split3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_split3.json').read().strip().splitlines()[-1])
print("split3:", d['value'], d['ms_per_step'], json.dumps({k:v.get('ms_per_image') for k,v in d['kernels'].items()}))
PY
python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider -k "tower_lanes" > gpurun_out/r06_call7_lanes.txt 2>&1; tail -2 gpurun_out/r06_call7_lanes.txt
MPN_FLAVOUR=debug python tools/hook_ab.py c3 12 base tower_order=0 > gpurun_out/r06_c3_order_ab.txt 2>&1; grep -v amdgpu gpurun_out/r06_c3_order_ab.txt
