#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -p no:cacheprovider -k "three_plane or stages_vs_oracle" > gpurun_out/r06_call4_split3_small.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call4_split3_small.txt
grep -v amdgpu gpurun_out/r06_call4_split3_small.txt | tail -8
python bench.py --fc-arith split3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_split3.json 2> gpurun_out/r06_bench_split3.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06_bench_split3.json').read().strip().splitlines()[-1])
    print("split3:", d['value'], d['ms_per_step'], json.dumps(d.get('fc6_split3')), json.dumps({k:v.get('ms_per_image') for k,v in d['kernels'].items()}))
except Exception as e:
    print("bench split3 failed", e); print(open('gpurun_out/r06_bench_split3.err').read()[-1500:])
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 --no-power-sensitivity > gpurun_out/r06_bench_fp32_ref.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_fp32_ref.json').read().strip().splitlines()[-1])
print("fp32 :", d['value'], d['ms_per_step'], json.dumps({k:v.get('ms_per_image') for k,v in d['kernels'].items()}))
PY
bash tools/r06_split3_gate.sh
