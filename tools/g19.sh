cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g19
for v in 0 1 2 8 16 32 7 23 24; do
  echo "=== variant: bf16_bdir=2 bf16_bdir_abl=$v"
  timeout 200 python tools/bench_conv_bf16.py all bf16_bdir=2 bf16_bdir_ver=8 bf16_bdir_abl=$v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g19/abl8.txt
tail -3 gpurun_out/g19/abl8.txt
