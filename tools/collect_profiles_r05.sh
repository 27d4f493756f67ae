# Round-5 evidence set (run on the GPU box: gpurun -- bash tools/collect_profiles_r05.sh).  Output: gpurun_out/prof_r05/ ; copy into profiles/.
TAG=r05
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the headline line (sustained leg, power_sensitivity, mixed-size leg, CPU baseline) + rocprof kernel stats of the same command
python $R/bench.py --mixed-sizes > $OUT/${TAG}_bench.json 2> $OUT/bench.err
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-power-sensitivity > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# 2. PMC passes of the headline config (separate passes; --kernel-trace only beside --pmc)
for set in "sq:GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${set%%:*}; ctrs=${set#*:}
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-power-sensitivity --sustained-seconds 0 > /tmp/pmc_$name.log 2>&1
  D=$(dirname $(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1))
  python $R/tools/pmc_summary.py $D > $OUT/${TAG}_pmc_$name.summary.txt 2>&1
done
# 3. the auxiliary lines: the headline model at 2000 proposals, the other BASELINE configs
timeout 600 python $R/bench.py --config c2 --rois 2000 --steps 10 --warmup 3 > $OUT/${TAG}_bench_c2_n2000.json 2> /tmp/bench_n2000.err
for cfg in "c1:--config c1 --steps 50 --warmup 6" "c3:--config c3 --steps 6 --warmup 2" "c4:--config c4 --steps 6 --warmup 2" "c4_bf16:--config c4 --dtype bf16 --steps 6 --warmup 2" "c5:--config c5 --steps 6 --warmup 2"; do
  key=${cfg%%:*}; args=${cfg#*:}
  timeout 900 python $R/bench.py $args > $OUT/${TAG}_bench_$key.json 2> /tmp/bench_$key.err
done
for cfg in "c5:--config c5" "c4_bf16:--config c4 --dtype bf16"; do
  key=${cfg%%:*}; args=${cfg#*:}
  rm -rf /tmp/kt_$key && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$key -o kt --output-format csv -- python $R/bench.py $args --steps 4 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > /tmp/kt_$key.out 2> /tmp/kt_$key.err
  cp $(find /tmp/kt_$key -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${key}_kernel_stats.csv
  for set in "sq:GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${set%%:*}; ctrs=${set#*:}
    rm -rf /tmp/pmc_${key}_$name
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${key}_$name -o p --output-format csv -- python $R/bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > /tmp/pmc_${key}_$name.log 2>&1
    D=$(dirname $(find /tmp/pmc_${key}_$name -name "*counter_collection.csv" | head -1))
    python $R/tools/pmc_summary.py $D > $OUT/${TAG}_${key}_pmc_$name.summary.txt 2>&1
  done
done
# 4. latency mode
timeout 600 python $R/bench.py --mode latency --steps 20 --warmup 5 > $OUT/${TAG}_bench_latency.json 2> /tmp/bench_lat.err
timeout 600 python $R/bench.py --mode latency --config c3 --steps 6 --warmup 2 > $OUT/${TAG}_bench_latency_c3.json 2> /tmp/bench_lat3.err
# 5. per-layer / per-path timings that the docs quote
timeout 300 python $R/tools/bench_layers.py > $OUT/${TAG}_vgg_layers.txt 2>&1
(timeout 300 python $R/tools/bench_nms.py 1000; timeout 300 python $R/tools/bench_nms.py 300) > $OUT/${TAG}_nms_paths.txt 2>&1
timeout 300 python $R/tools/nms_fused_trace.py > $OUT/${TAG}_nms_fused_trace.txt 2>&1
timeout 300 python $R/tools/ablate_wino_prologue.py > $OUT/${TAG}_wino_prologue_ceiling.txt 2>&1
(cd $R && timeout 300 python -m pytest tests/test_gpu_nms.py -k dropin_cost -m gpu -q -s -p no:cacheprovider > $OUT/${TAG}_libnms_dropin.txt 2>&1)
# 6. five back-to-back headline runs (spread between runs on one box)
for i in 1 2 3 4 5; do python $R/bench.py --no-cpu-baseline --no-power-sensitivity --sustained-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'])"; done > $OUT/${TAG}_bench_repeats.txt 2>&1
ls -la $OUT
