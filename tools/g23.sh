cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g23
for v in "bf16_bdir=0" "bf16_bdir=2" "bf16_bdir=1"; do
  echo "=== variant: $v"
  timeout 300 python tools/bench_conv_bf16.py all --clk $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g23/rag.txt
cut -c1-60,100-150 gpurun_out/g23/rag.txt
timeout 600 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_inception.py -q -x -m gpu 2>&1 | tail -3
