cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g27
for v in "bf16_bdir=2 bf16_bdir_abl=256" "bf16_bdir=2" "bf16_bdir=1"; do
  echo "=== variant: $v"
  timeout 300 python tools/bench_conv_bf16.py all --clk $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g27/fine.txt
cut -c1-60,100-175 gpurun_out/g27/fine.txt
