cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g32
for v in "bf16_bdir=0" "bf16_bdir=0 bf16_dma_tn=2256" "bf16_bdir=2"; do
  echo "=== variant: $v"
  timeout 300 python tools/bench_conv_bf16.py all --clk $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g32/w8.txt
grep -E "variant|tower" gpurun_out/g32/w8.txt
