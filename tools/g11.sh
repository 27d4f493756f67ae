cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g11
timeout 1200 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_inception.py tests/test_gpu_fullsize_graphs.py -m gpu -q > gpurun_out/g11/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g11/tests.log
tail -n 8 gpurun_out/g11/tests.log
timeout 1200 python -m pytest tests/test_gpu_graphs_rigor.py -m gpu -q -k "bf16" > gpurun_out/g11/rigor.log 2>&1
echo "rigor rc=$?" >> gpurun_out/g11/rigor.log
tail -n 5 gpurun_out/g11/rigor.log
for c in "c4 --dtype bf16" "c5"; do timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > gpurun_out/g11/bench_$(echo $c | tr ' -' '__').json 2>/dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/g11/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"]["fc6"]["ms_per_image"])
PY
