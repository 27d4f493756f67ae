cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g21
for v in 0 64 192 24 23; do
  echo "=== variant: bf16_bdir=2 bf16_bdir_abl=$v"
  timeout 300 python tools/bench_conv_bf16.py inception --clk bf16_bdir=2 bf16_bdir_abl=$v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g21/clk.txt
cut -c1-60,100-200 gpurun_out/g21/clk.txt
