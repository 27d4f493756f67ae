# Round-5 addendum: the fused NMS kernel with the lazy replay on its LDS mask, dispatched up to 1024 rows outside the pipelined forms.
# Re-measures only what that touches (gpurun -- bash tools/collect_nms_r05b.sh).  Output: gpurun_out/prof_r05b/ ; copy into profiles/.
TAG=r05b
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --mode latency --steps 20 --warmup 5 > $OUT/${TAG}_bench_latency.json 2> /tmp/bench_lat.err
timeout 600 python $R/bench.py --mode latency --config c3 --steps 6 --warmup 2 > $OUT/${TAG}_bench_latency_c3.json 2> /tmp/bench_lat3.err
(timeout 300 python $R/tools/bench_nms.py 1000; timeout 300 python $R/tools/bench_nms.py 300; MPN_FUSED_REPLAY=0 timeout 120 python $R/tools/bench_nms.py 1000 fewties 5; MPN_FUSED_REPLAY=0 timeout 120 python $R/tools/bench_nms.py 300 fewties 5) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_nms_paths.txt
timeout 300 python $R/tools/nms_fused_trace.py > $OUT/${TAG}_nms_fused_trace.txt 2>&1
(cd $R && timeout 300 python -m pytest tests/test_gpu_nms.py -k dropin_cost -m gpu -q -s -p no:cacheprovider > $OUT/${TAG}_libnms_dropin.txt 2>&1)
python $R/bench.py --no-cpu-baseline --no-power-sensitivity --sustained-seconds 0 > $OUT/${TAG}_bench_headline_check.json 2>/dev/null
timeout 600 python $R/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_c5.json 2> /tmp/bench_c5.err
ls -la $OUT
