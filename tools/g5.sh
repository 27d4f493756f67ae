cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
for v in "bf16_exp=22" "bf16_exp=54" "bf16_exp=86" "bf16_exp=118" ; do
  echo "=== variant: $v" >> gpurun_out/g5/conv.log
  timeout 300 python tools/bench_conv_bf16.py all $v >> gpurun_out/g5/conv.log 2>&1
done
for v in "" "bf16_exp=22" "bf16_exp=118"; do
  echo "=== variant: ZERO-FILL $v" >> gpurun_out/g5/conv.log
  MPN_BENCH_FILL=zero timeout 300 python tools/bench_conv_bf16.py all $v >> gpurun_out/g5/conv.log 2>&1
done
grep -E "tower conv|variant" gpurun_out/g5/conv.log
