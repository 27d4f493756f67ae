#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -p no:cacheprovider -k "three_plane or row_scales_in_place or tower_lanes or multipathnet" > gpurun_out/r06_call5_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call5_tests.txt
grep -v amdgpu gpurun_out/r06_call5_tests.txt | grep "passed\|failed\|max |d|\|float64\|rc " | tail -8
bash tools/r06_split3_gate.sh
python bench.py --fc-arith split3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_split3.json 2> gpurun_out/r06_bench_split3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_split3.json').read().strip().splitlines()[-1])
print("split3:", d['value'], d['ms_per_step'], json.dumps({k:v.get('ms_per_image') for k,v in d['kernels'].items()}))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt3 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fc-arith split3 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > /tmp/kt3.out 2> /tmp/kt3.err
cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r06_split3_kernel_stats.csv; head -12 $GRAFT_REPO_ROOT/gpurun_out/r06_split3_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
MPN_FLAVOUR=debug python tools/hook_ab.py c3 12 base gemm_rsi=0 tower_lanes=0 tower_lanes=0,tower_share=0,gemm_rsi=0 > gpurun_out/r06_c3_ab.txt 2>&1; grep -v amdgpu gpurun_out/r06_c3_ab.txt
python bench.py --config c3 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c3_b.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_c3_b.json').read().strip().splitlines()[-1])
print("c3:", d['value'], d['ms_per_step'], d['roofline'].get('frac'), json.dumps({k:v.get('ms_per_image') for k,v in d['kernels'].items()}))
PY
