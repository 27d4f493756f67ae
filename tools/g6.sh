cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g6
for v in "" "bf16_exp=22" "bf16_exp=118" "bf16_exp=8"; do
  echo "=== variant: $v" >> gpurun_out/g6/conv.log
  timeout 600 python tools/bench_conv_bf16.py all --clk $v >> gpurun_out/g6/conv.log 2>&1
done
grep -v amdgpu gpurun_out/g6/conv.log
