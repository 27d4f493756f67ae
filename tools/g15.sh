cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g15
for v in "bf16_bdir=0" "bf16_bdir=2" "bf16_bdir=2 bf16_bdir_ver=2" "bf16_bdir=2 bf16_bdir_ver=3"; do
  echo "=== variant: $v" >> gpurun_out/g15/conv.log
  timeout 300 python tools/bench_conv_bf16.py all $v >> gpurun_out/g15/conv.log 2>&1
done
grep -v amdgpu gpurun_out/g15/conv.log | cut -c1-75,100-140
