cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g10
for v in "" "bf16_bdir=3" "bf16_bdir=4"; do
  echo "=== variant: $v" >> gpurun_out/g10/conv.log
  timeout 300 python tools/bench_conv_bf16.py all $v >> gpurun_out/g10/conv.log 2>&1
done
grep -v amdgpu gpurun_out/g10/conv.log
