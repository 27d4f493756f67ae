cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
ls -la /sys/class/drm/ > gpurun_out/g2/sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo "== $d"; ls $d/hwmon/* 2>&1 | head -40; cat $d/pp_dpm_sclk 2>&1 | head; done >> gpurun_out/g2/sysfs.txt 2>&1
(timeout 20 rocm-smi --showclocks --showpower --json; echo; timeout 20 amd-smi metric --json 2>&1 | head -80) >> gpurun_out/g2/sysfs.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_graphs_rigor.py tests/test_gpu_shard.py -m gpu -q -s > gpurun_out/g2/rigor.log 2>&1
echo "rigor rc=$?" >> gpurun_out/g2/rigor.log
timeout 600 python bench.py > gpurun_out/g2/bench.json 2> gpurun_out/g2/bench.err
for c in c1 c3 c4 c5; do timeout 600 python bench.py --config $c --steps $([ $c = c1 ] && echo 50 || echo 6) --warmup 2 > gpurun_out/g2/bench_$c.json 2> gpurun_out/g2/bench_$c.err; done
timeout 300 python bench.py --config c4 --dtype bf16 --steps 6 --warmup 2 > gpurun_out/g2/bench_c4_bf16.json 2> gpurun_out/g2/bench_c4b.err
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_graphs_rigor.py --deselect tests/test_gpu_shard.py > gpurun_out/g2/suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/g2/suite.log
tail -5 gpurun_out/g2/rigor.log gpurun_out/g2/suite.log
