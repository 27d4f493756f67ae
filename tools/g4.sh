cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
for v in "" "bf16_dma_tn=256" "bf16_dma_tn=128" "bf16_dma_tn=1256" "bf16_exp=2" "bf16_exp=4" "bf16_exp=6" "bf16_exp=8" "bf16_exp=16" "bf16_exp=24" "bf16_exp=22"; do
  echo "=== variant: $v" >> gpurun_out/g4/conv.log
  timeout 300 python tools/bench_conv_bf16.py all $v >> gpurun_out/g4/conv.log 2>&1
done
grep -E "tower conv|variant" gpurun_out/g4/conv.log
