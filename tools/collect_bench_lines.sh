# The bench.py lines of the evidence set only (headline + rocprof kernel stats of the same command, the other BASELINE configs, the
# latency mode), for a re-collection after a change that does not touch the kernels the per-kernel traces describe.
# gpurun -- bash tools/collect_bench_lines.sh r03   -> gpurun_out/prof_<tag>/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
for c in c1 c3 c4 c5; do
  timeout 600 python $R/bench.py --config $c --steps $([ $c = c1 ] && echo 50 || echo 6) --warmup 2 > $OUT/${TAG}_bench_$c.json 2> /tmp/bench_$c.err
done
timeout 600 python $R/bench.py --config c4 --dtype bf16 --steps 6 --warmup 2 > $OUT/${TAG}_bench_c4_bf16.json 2> /tmp/bench_c4b.err
timeout 600 python $R/bench.py --mode latency --steps 20 --warmup 5 > $OUT/${TAG}_bench_latency.json 2> /tmp/bench_lat.err
timeout 600 python $R/bench.py --mode latency --config c3 --steps 6 --warmup 2 > $OUT/${TAG}_bench_latency_c3.json 2> /tmp/bench_lat3.err
# configs[0] in detail: per-launch timeline of one image, rocprof kernel stats, host enqueue rate vs device rate
bash $R/tools/kernel_timeline.sh c1 > $OUT/${TAG}_alexnet_timeline.txt 2>&1
rm -rf /tmp/kt_c1 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_c1 -o kt --output-format csv -- python $R/bench.py --config c1 --no-cpu-baseline > /tmp/kt_c1.out 2> /tmp/kt_c1.err
cp $(find /tmp/kt_c1 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_alexnet_kernel_stats.csv
(cd $R && timeout 300 python tools/host_enqueue_probe.py c1 > $OUT/${TAG}_host_enqueue_probe.txt 2>&1; timeout 300 python tools/host_enqueue_probe.py c2 50 >> $OUT/${TAG}_host_enqueue_probe.txt 2>&1)
for i in 1 2 3 4 5; do python $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done > $OUT/${TAG}_bench_repeats.txt
ls -la $OUT
