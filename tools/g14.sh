cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g14
timeout 300 python tools/bench_conv_bf16.py all > gpurun_out/g14/conv_rule.log 2>&1
grep "tower conv" gpurun_out/g14/conv_rule.log
for c in "c4 --dtype bf16" "c5"; do timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > gpurun_out/g14/bench_$(echo $c | tr ' -' '__').json 2>/dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/g14/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"]["fc6"]["ms_per_image"])
PY
