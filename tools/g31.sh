cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g31
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_inception.py tests/test_gpu_fullsize_graphs.py -q -x -m gpu 2>&1 | tail -3
for v in "bf16_bdir=1"; do
  echo "=== variant: $v"
  timeout 300 python tools/bench_conv_bf16.py all --clk $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g31/cvt.txt
grep -E "variant|tower" gpurun_out/g31/cvt.txt
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g31/c5.json 2> gpurun_out/g31/c5.err
python bench.py --config c4 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g31/c4b.json 2> gpurun_out/g31/c4b.err
python - <<'PY'
import json
for f in ('c5','c4b'):
    try:
        d=json.loads(open('gpurun_out/g31/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('sustained',{}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
