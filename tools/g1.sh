cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
timeout 1500 python -m pytest tests/test_gpu_graphs_rigor.py tests/test_gpu_shard.py -m gpu -x -q -s > gpurun_out/g1/rigor.log 2>&1
echo "rigor rc=$?" >> gpurun_out/g1/rigor.log
for c in c4 c5; do timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g1/bench_$c.json 2> gpurun_out/g1/bench_$c.err; done
timeout 300 python bench.py --config c4 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g1/bench_c4_bf16.json 2> gpurun_out/g1/bench_c4b.err
timeout 300 python bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g1/bench_c3.json 2> gpurun_out/g1/bench_c3.err
tail -5 gpurun_out/g1/rigor.log
