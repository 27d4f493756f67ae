cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
timeout 1200 python -m pytest tests/test_gpu_launch_graphs.py tests/test_gpu_shard.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/g8/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g8/tests.log
tail -n 12 gpurun_out/g8/tests.log
for g in 1 0; do
  MPN_GRAPHS=$g timeout 300 python bench.py --no-cpu-baseline > gpurun_out/g8/bench_g$g.json 2> gpurun_out/g8/bench_g$g.err
  MPN_GRAPHS=$g timeout 300 python bench.py --config c1 --steps 50 --warmup 6 --no-cpu-baseline > gpurun_out/g8/bench_c1_g$g.json 2> gpurun_out/g8/bench_c1_g$g.err
  MPN_GRAPHS=$g timeout 300 python bench.py --mode latency --steps 20 --warmup 5 > gpurun_out/g8/bench_lat_g$g.json 2> gpurun_out/g8/bench_lat_g$g.err
  MPN_GRAPHS=$g timeout 300 python tools/host_enqueue_probe.py c1 > gpurun_out/g8/enqueue_c1_g$g.txt 2>&1
done
python - <<'PY'
import json
for g in (1, 0):
    for f in ("bench", "bench_c1", "bench_lat"):
        try:
            d = json.loads(open("gpurun_out/g8/%s_g%d.json" % (f, g)).read().strip().splitlines()[-1])
            print(f, "graphs", g, d["value"], d["ms_per_step"], d.get("unsharded_ms"), (d.get("projected") or {}).get("rank0_compute_ms"), (d.get("sustained") or {}).get("value"))
        except Exception as e:
            print(f, g, "ERR", e)
PY
tail -n 5 gpurun_out/g8/enqueue_c1_g1.txt gpurun_out/g8/enqueue_c1_g0.txt
