cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g30
for v in "bf16_bdir=2" "bf16_bdir=2 bf16_ty_group=2" "bf16_bdir=2 bf16_ty_group=4" "bf16_bdir=2 bf16_ty_group=6" "bf16_bdir=2 bf16_ty_group=1"; do
  echo "=== variant: $v"
  timeout 300 python tools/bench_conv_bf16.py all --clk $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g30/grp.txt
grep -E "variant|tower" gpurun_out/g30/grp.txt
