#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_alexnet.py tests/test_gpu_resnet.py tests/test_gpu_inception.py tests/test_gpu_shard.py tests/test_gpu_launch_graphs.py tests/test_gpu_roipool.py -m gpu -q -p no:cacheprovider > gpurun_out/r06_call3_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call3_tests.txt
tail -6 gpurun_out/r06_call3_tests.txt
MPN_FLAVOUR=debug python tools/tower_lanes_ab.py c3 12 > gpurun_out/r06_tower_lanes_ab_c3.txt 2>&1; grep -v amdgpu gpurun_out/r06_tower_lanes_ab_c3.txt | tail -5
MPN_FLAVOUR=debug python tools/hook_ab.py c5 12 base pool_exp=1 pool_exp=2 pool_exp=3 > gpurun_out/r06_pool_exp.txt 2>&1
MPN_FLAVOUR=debug python tools/hook_ab.py c4 12 base pool_exp=1 pool_exp=2 pool_exp=3 >> gpurun_out/r06_pool_exp.txt 2>&1
grep -v amdgpu gpurun_out/r06_pool_exp.txt
python bench.py --config c3 --steps 8 --warmup 3 > gpurun_out/r06_bench_c3_a.json 2> gpurun_out/r06_bench_c3_a.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_c3_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('frac'), json.dumps(d.get('groups'))[:600])
PY
