cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g18
for v in "bf16_bdir=2" "bf16_bdir=2 bf16_bdir_ver=8"; do
  echo "=== variant: $v"
  timeout 200 python tools/bench_conv_bf16.py all $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g18/b8.txt
cat gpurun_out/g18/b8.txt | cut -c1-150
