# Round-6 evidence set (run on the GPU box: gpurun -- bash tools/collect_profiles_r06.sh).  Output: gpurun_out/prof_r06/ ; copy into profiles/, then
# (here, where .git is) python tools/traffic_from_pmc.py r06 [--config ...] regenerates profiles/traffic.json and stamps the commit.
TAG=r06
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
# 1. the headline line (sustained leg, power_sensitivity, mixed-size leg, CPU baseline) + rocprof kernel stats of the same command
python $R/bench.py --mixed-sizes > $OUT/${TAG}_bench.json 2> $OUT/bench.err
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-power-sensitivity > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# 2. PMC passes of the headline config (separate passes; --kernel-trace only beside --pmc)
for set in "sq:$SQ" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${set%%:*}; ctrs=${set#*:}
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-power-sensitivity --sustained-seconds 0 > /tmp/pmc_$name.log 2>&1
  D=$(dirname $(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1))
  python $R/tools/pmc_summary.py $D > $OUT/${TAG}_pmc_$name.summary.txt 2>&1
done
# 3. the auxiliary lines: the headline model at 2000 proposals, fc6 / fc7 as the three-plane bf16 split, the other BASELINE configs
timeout 600 python $R/bench.py --config c2 --rois 2000 --steps 10 --warmup 3 > $OUT/${TAG}_bench_c2_n2000.json 2> /tmp/bench_n2000.err
timeout 600 python $R/bench.py --fc-arith split3 --steps 20 --warmup 5 > $OUT/${TAG}_bench_split3.json 2> /tmp/bench_split3.err
timeout 600 python $R/bench.py --fc-arith split3 --rois 2000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_split3_n2000.json 2> /tmp/bench_split3b.err
timeout 600 python $R/bench.py --config c3 --fc-arith split3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_c3_split3.json 2> /tmp/bench_split3c.err
for cfg in "c1:--config c1 --steps 50 --warmup 6" "c3:--config c3 --steps 20 --warmup 5" "c4:--config c4 --steps 6 --warmup 2" "c4_bf16:--config c4 --dtype bf16 --steps 6 --warmup 2" "c5:--config c5 --steps 6 --warmup 2"; do
  key=${cfg%%:*}; args=${cfg#*:}
  timeout 900 python $R/bench.py $args > $OUT/${TAG}_bench_$key.json 2> /tmp/bench_$key.err
done
# 4. kernel stats + PMC of the split3 line and of the three tower configs
rm -rf /tmp/kt_s3 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_s3 -o kt --output-format csv -- python $R/bench.py --fc-arith split3 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > /tmp/kt_s3.out 2> /tmp/kt_s3.err
cp $(find /tmp/kt_s3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_split3_kernel_stats.csv
rm -rf /tmp/pmc_s3 && timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d /tmp/pmc_s3 -o p --output-format csv -- python $R/bench.py --fc-arith split3 --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > /tmp/pmc_s3.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/pmc_s3 -name "*counter_collection.csv" | head -1)) > $OUT/${TAG}_split3_pmc_sq.summary.txt 2>&1
for cfg in "c3:--config c3" "c5:--config c5" "c4_bf16:--config c4 --dtype bf16"; do
  key=${cfg%%:*}; args=${cfg#*:}
  rm -rf /tmp/kt_$key && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$key -o kt --output-format csv -- python $R/bench.py $args --steps 4 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > /tmp/kt_$key.out 2> /tmp/kt_$key.err
  cp $(find /tmp/kt_$key -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${key}_kernel_stats.csv
  for set in "sq:$SQ" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${set%%:*}; ctrs=${set#*:}
    rm -rf /tmp/pmc_${key}_$name
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${key}_$name -o p --output-format csv -- python $R/bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > /tmp/pmc_${key}_$name.log 2>&1
    D=$(dirname $(find /tmp/pmc_${key}_$name -name "*counter_collection.csv" | head -1))
    python $R/tools/pmc_summary.py $D > $OUT/${TAG}_${key}_pmc_$name.summary.txt 2>&1
  done
done
# 5. latency mode
timeout 600 python $R/bench.py --mode latency --steps 20 --warmup 5 > $OUT/${TAG}_bench_latency.json 2> /tmp/bench_lat.err
timeout 600 python $R/bench.py --mode latency --config c3 --steps 6 --warmup 2 > $OUT/${TAG}_bench_latency_c3.json 2> /tmp/bench_lat3.err
# 6. five back-to-back headline runs (spread between runs on one box)
for i in 1 2 3 4 5; do python $R/bench.py --no-cpu-baseline --no-power-sensitivity --sustained-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'])"; done > $OUT/${TAG}_bench_repeats.txt 2>&1
ls -la $OUT
