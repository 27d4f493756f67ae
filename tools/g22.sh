cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g22
for v in 198 194 70 6 7 2; do
  echo "=== variant: bf16_bdir=2 bf16_bdir_abl=$v"
  timeout 300 python tools/bench_conv_bf16.py inception --clk bf16_bdir=2 bf16_bdir_abl=$v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g22/clk.txt
grep -E "variant|tower" gpurun_out/g22/clk.txt
