cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
timeout 1500 python -m pytest tests/test_gpu_graphs_rigor.py -m gpu -q -s -k "mpn" > gpurun_out/g3/rigor.log 2>&1
echo "rigor rc=$?" >> gpurun_out/g3/rigor.log
for v in "" "bf16_exp=1" "bf16_nch=8" "bf16_nch=8 bf16_exp=1"; do
  echo "=== variant: $v" >> gpurun_out/g3/conv.log
  timeout 300 python tools/bench_conv_bf16.py all $v >> gpurun_out/g3/conv.log 2>&1
done
grep -E "tower conv|variant" gpurun_out/g3/conv.log
tail -n 3 gpurun_out/g3/rigor.log
